// udpbroadcast.cpp -- see udpbroadcast.h.
#include "udpbroadcast.h"

#include <arpa/inet.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

namespace udpbroadcast {

namespace detail {

DatagramSocket::DatagramSocket() : fd_(::socket(AF_INET, SOCK_DGRAM, 0))
{
    if (fd_ < 0) throw "error sock";
}

DatagramSocket::~DatagramSocket()
{
    if (fd_ >= 0) ::close(fd_);
}

sockaddr_in DatagramSocket::address(unsigned long host_order_ip, int port)
{
    sockaddr_in a;
    memset(&a, 0, sizeof a);
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl((uint32_t)host_order_ip);
    a.sin_port = htons((uint16_t)port);
    return a;
}

}   // namespace detail

udpclient::udpclient(int port) : to_(detail::DatagramSocket::address(INADDR_BROADCAST, port))
{
    const int enable = 1;
    ::setsockopt(sock_.fd(), SOL_SOCKET, SO_BROADCAST, &enable, sizeof enable);
}

udpclient::udpclient(int port, const char *ipv4) : to_(detail::DatagramSocket::address(INADDR_BROADCAST, port))
{
    if (!ipv4 || ::inet_pton(AF_INET, ipv4, &to_.sin_addr) != 1) throw "error address";
}

udpclient::~udpclient() {}

int udpclient::send(const char *message, size_t length)
{
    return (int)::sendto(sock_.fd(), message, length, 0, reinterpret_cast<const sockaddr *>(&to_), sizeof to_);
}

udpserver::udpserver(int port)
{
    memset(&from_, 0, sizeof from_);
    // a sector arrives as a burst of m datagrams (6 MiB): ask for as much kernel buffer as the host grants
    const int want = 16 << 20;
    ::setsockopt(sock_.fd(), SOL_SOCKET, SO_RCVBUF, &want, sizeof want);
    const sockaddr_in any = detail::DatagramSocket::address(INADDR_ANY, port);
    if (::bind(sock_.fd(), reinterpret_cast<const sockaddr *>(&any), sizeof any) != 0) throw "error bind";
}

udpserver::~udpserver() {}

void udpserver::set_timeout_ms(int ms)
{
    timeval tv;
    tv.tv_sec = ms / 1000;
    tv.tv_usec = (ms % 1000) * 1000;
    ::setsockopt(sock_.fd(), SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
}

int udpserver::recv(char *buffer, size_t length)
{
    socklen_t alen = sizeof from_;
    return (int)::recvfrom(sock_.fd(), buffer, length, MSG_WAITALL, reinterpret_cast<sockaddr *>(&from_), &alen);
}

}   // namespace udpbroadcast
