// udpbroadcast.h -- datagram endpoints for the radar's ingest (port 19001) and product egress
// (19002 Zdb, 19003 Zdr).  Transport only; nothing here is on the accelerated path.
//
// namespace, class names, constructors and send/recv signatures are those a caller of the
// reference's udpbroadcast.h:8-30 uses; construction failures throw a const char* as there
// (udpbroadcast.cpp:19,57).  Both endpoints share one RAII socket holder.
#ifndef WRP_HOST_UDPBROADCAST_H
#define WRP_HOST_UDPBROADCAST_H

#include <netinet/in.h>
#include <stddef.h>

namespace udpbroadcast {

namespace detail {
// owns one IPv4 datagram socket; closes it on destruction; not copyable
class DatagramSocket {
  public:
    DatagramSocket();                       // throws "error sock"
    ~DatagramSocket();
    DatagramSocket(const DatagramSocket &) = delete;
    DatagramSocket &operator=(const DatagramSocket &) = delete;
    int fd() const { return fd_; }
    static sockaddr_in address(unsigned long host_order_ip, int port);
  private:
    int fd_;
};
}   // namespace detail

// sends every datagram to the limited broadcast address 255.255.255.255:port
class udpclient {
  public:
    udpclient(int port);
    // extension (not in the reference): unicast to a dotted-quad IPv4 address, e.g. "127.0.0.1" on a
    // host without a broadcast route; throws "error address" if it does not parse
    udpclient(int port, const char *ipv4);
    ~udpclient();
    int send(const char *message, size_t length);   // bytes sent or -1
  private:
    detail::DatagramSocket sock_;
    sockaddr_in to_;
};

// bound to 0.0.0.0:port; recv blocks until one datagram (at most `length` bytes) arrives
class udpserver {
  public:
    udpserver(int port);                            // throws "error bind"
    ~udpserver();
    int recv(char *buffer, size_t length);          // bytes received or -1
    // extension (not in the reference): recv gives up after `ms` milliseconds without a datagram (-1, errno EAGAIN),
    // so that a reader can look at a stop flag; 0 = block for ever (the default)
    void set_timeout_ms(int ms);
  private:
    detail::DatagramSocket sock_;
    sockaddr_in from_;
};

}   // namespace udpbroadcast
#endif   // WRP_HOST_UDPBROADCAST_H
