// tcp.cpp -- see tcp.h.
#include "tcp.h"

#include <arpa/inet.h>
#include <errno.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cstring>
#include <vector>

namespace tcp {

namespace detail {

Descriptor::~Descriptor()
{
    if (fd_ >= 0) ::close(fd_);
}

Descriptor &Descriptor::operator=(Descriptor &&o) noexcept
{
    if (this != &o) {
        if (fd_ >= 0) ::close(fd_);
        fd_ = o.fd_;
        o.fd_ = -1;
    }
    return *this;
}

int read_all(int fd, char *buffer, size_t length)
{
    size_t done = 0;
    while (done < length) {
        const ssize_t n = ::read(fd, buffer + done, length - done);
        if (n > 0) done += (size_t)n;
        else if (n == 0) return -1;                 // orderly shutdown before the message was complete
        else if (errno != EINTR) return -1;
    }
    return (int)done;
}

int write_all(int fd, const char *buffer, size_t length)
{
    size_t done = 0;
    while (done < length) {
        const ssize_t n = ::send(fd, buffer + done, length - done, MSG_NOSIGNAL);
        if (n > 0) done += (size_t)n;
        else if (n < 0 && errno != EINTR) return -1;
    }
    return (int)done;
}

static sockaddr_in endpoint(unsigned long host_order_ip, int port)
{
    sockaddr_in a;
    std::memset(&a, 0, sizeof a);
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(host_order_ip);
    a.sin_port = htons((unsigned short)port);
    return a;
}

}   // namespace detail

tcpclient::tcpclient(int port) : sock_(::socket(AF_INET, SOCK_STREAM, 0))
{
    if (sock_.get() < 0) throw "error sock";
    const sockaddr_in to = detail::endpoint(INADDR_LOOPBACK, port);   // the reference's fixed 127.0.0.1 (tcp.cpp:26)
    if (::connect(sock_.get(), reinterpret_cast<const sockaddr *>(&to), sizeof to) != 0) throw "error connect";
}

tcpclient::~tcpclient() {}

int tcpclient::sendit(const char *message, size_t length)
{
    if (detail::write_all(sock_.get(), message, length) < 0) return -1;
    std::vector<char> ack(length);                  // the echo is the acknowledgement (tcp.cpp:45-50)
    return detail::read_all(sock_.get(), ack.data(), length) < 0 ? -1 : 0;
}

tcpserver::tcpserver(int port) : listen_(::socket(AF_INET, SOCK_STREAM, 0))
{
    if (listen_.get() < 0) throw "error sock";
    int on = 1;
    if (::setsockopt(listen_.get(), SOL_SOCKET, SO_REUSEADDR, &on, sizeof on) != 0) throw "error sock opt";
    const sockaddr_in any = detail::endpoint(INADDR_ANY, port);
    if (::bind(listen_.get(), reinterpret_cast<const sockaddr *>(&any), sizeof any) != 0) throw "error bind";
    if (::listen(listen_.get(), 3) != 0) throw "error listen";
    int fd;
    do fd = ::accept(listen_.get(), nullptr, nullptr); while (fd < 0 && errno == EINTR);
    if (fd < 0) throw "error accept";
    peer_ = detail::Descriptor(fd);
}

tcpserver::~tcpserver() {}

int tcpserver::recv(char *buffer, size_t length)
{
    const int n = detail::read_all(peer_.get(), buffer, length);
    if (n < 0) return -1;
    if (detail::write_all(peer_.get(), buffer, length) < 0) return -1;
    return n;
}

}   // namespace tcp
