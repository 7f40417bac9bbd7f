// udpbroadcast.h -- UDP broadcast endpoints, API-compatible with the reference
// (udpbroadcast.h:8-30).  I/O only; not part of the accelerated path.
#ifndef WRP_HOST_UDPBROADCAST_H
#define WRP_HOST_UDPBROADCAST_H
#include <netinet/in.h>
#include <stddef.h>

namespace udpbroadcast {

class udpclient {
  private:
    int mPort;
    int sockfd;
    struct sockaddr_in servaddr;
  public:
    udpclient(int);                 // throws const char* like the reference (udpbroadcast.cpp:19)
    ~udpclient();
    int send(const char *, size_t);
};

class udpserver {
  private:
    int mPort;
    int sockfd;
    struct sockaddr_in servaddr, cliaddr;
  public:
    udpserver(int);
    ~udpserver();
    int recv(char *, size_t);
};

} // namespace udpbroadcast
#endif
