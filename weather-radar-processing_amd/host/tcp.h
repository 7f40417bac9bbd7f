// tcp.h -- stream endpoints with the reference's names and signatures (tcp.h:8-30 there): a client
// that connects to 127.0.0.1:port and whose sendit() waits for the peer to echo the message back as
// the acknowledgement, and a server that accepts ONE peer in its constructor and whose recv() reads a
// message and echoes it.  Transport only; nothing here is on the accelerated path, and like the
// reference's class no binary uses it (compiled by `make all`, used by no main).
//
// Construction failures throw a const char* as there (tcp.cpp:17-31,62-90).  Deliberate differences,
// both fixes of defects a caller could only trip over: the server talks on the ACCEPTED connection
// (the reference reads and writes its listening descriptor, tcp.cpp:96-100, which fails with ENOTCONN),
// and partial reads/writes are completed instead of being reported as the whole message.
#ifndef WRP_HOST_TCP_H
#define WRP_HOST_TCP_H

#include <stddef.h>

namespace tcp {

namespace detail {
// owns one descriptor; closes it on destruction; movable, not copyable
class Descriptor {
  public:
    explicit Descriptor(int fd = -1) : fd_(fd) {}
    ~Descriptor();
    Descriptor(const Descriptor &) = delete;
    Descriptor &operator=(const Descriptor &) = delete;
    Descriptor(Descriptor &&o) noexcept : fd_(o.fd_) { o.fd_ = -1; }
    Descriptor &operator=(Descriptor &&o) noexcept;
    int get() const { return fd_; }
  private:
    int fd_;
};
// all `length` bytes or -1 (peer gone / error); retries on EINTR and short transfers
int read_all(int fd, char *buffer, size_t length);
int write_all(int fd, const char *buffer, size_t length);
}   // namespace detail

class tcpclient {
  public:
    tcpclient(int port);                              // throws "error sock" / "error connect"
    ~tcpclient();
    int sendit(const char *message, size_t length);   // 0 = sent and acknowledged, -1 = peer gone
  private:
    detail::Descriptor sock_;
};

class tcpserver {
  public:
    tcpserver(int port);                              // blocks until one peer connects; throws "error bind" ...
    ~tcpserver();
    int recv(char *buffer, size_t length);            // bytes received (== length) or -1; echoes them
  private:
    detail::Descriptor listen_, peer_;
};

}   // namespace tcp
#endif   // WRP_HOST_TCP_H
