// udpbroadcast.cpp -- thin BSD-socket wrappers with the reference's behaviour
// (udpbroadcast.cpp:15-71): client sends to INADDR_BROADCAST:port, server binds ANY:port,
// failures throw a const char*.
#include "udpbroadcast.h"

#include <arpa/inet.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <unistd.h>

namespace udpbroadcast {

udpclient::udpclient(int port) : mPort(port)
{
    sockfd = socket(AF_INET, SOCK_DGRAM, 0);
    if (sockfd < 0) throw "error sock";
    int on = 1;
    setsockopt(sockfd, SOL_SOCKET, SO_BROADCAST, &on, sizeof(on));
    memset(&servaddr, 0, sizeof(servaddr));
    servaddr.sin_family = AF_INET;
    servaddr.sin_port = htons((unsigned short)mPort);
    servaddr.sin_addr.s_addr = htonl(INADDR_BROADCAST);
}

udpclient::~udpclient() { close(sockfd); }

int udpclient::send(const char *message, size_t length)
{
    return (int)sendto(sockfd, message, length, 0, (const struct sockaddr *)&servaddr, sizeof(servaddr));
}

udpserver::udpserver(int port) : mPort(port)
{
    sockfd = socket(AF_INET, SOCK_DGRAM, 0);
    if (sockfd < 0) throw "error sock";
    memset(&servaddr, 0, sizeof(servaddr));
    memset(&cliaddr, 0, sizeof(cliaddr));
    servaddr.sin_family = AF_INET;
    servaddr.sin_addr.s_addr = htonl(INADDR_ANY);
    servaddr.sin_port = htons((unsigned short)mPort);
    if (bind(sockfd, (const struct sockaddr *)&servaddr, sizeof(servaddr)) < 0) {
        close(sockfd);
        throw "error bind";
    }
}

udpserver::~udpserver() { close(sockfd); }

int udpserver::recv(char *buffer, size_t length)
{
    socklen_t len = sizeof(cliaddr);
    return (int)recvfrom(sockfd, buffer, length, MSG_WAITALL, (struct sockaddr *)&cliaddr, &len);
}

} // namespace udpbroadcast
