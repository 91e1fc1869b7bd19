// TEST INFRASTRUCTURE — a host-staged stand-in for the slice of the RCCL API that libatx binds (atx_comm.hip), so that the
// multi-peer paths of atx_exchange / atx_gather_shards / atx_bcast run with several ranks on a box with ONE GPU
// (tests/test_gpu_rccl_stub.py; selected through ATX_RCCL_LIBRARY).  Not a collective library: every call synchronises the
// stream, stages through the host and moves bytes over Unix-domain sockets.  What it checks is the caller's bookkeeping — who
// sends how many bytes to whom, in which order, into which buffer — which is exactly what the real library would be handed.
//
// Protocol: rank r listens on /tmp/atx-rccl-stub-<token>-<r>; a message is {int32 source, int64 bytes} + payload; a receiver
// thread drains every connection into a per-source FIFO, so a send never waits for the matching receive (no deadlock whatever the
// order inside a group).  Inside ncclGroupStart/End operations are deferred and run sends first, then receives.
#include <hip/hip_runtime.h>

#include "../../anemoi-transform_amd/csrc/atx_nccl_abi.h"  // NCCL's public C types: the stand-in is defined with the very signatures libatx calls through
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Comm {
    int world = 0, rank = 0;
    std::string token;
    int listen_fd = -1;
    std::thread acceptor;
    std::vector<std::thread> readers;
    std::mutex mu;
    std::condition_variable cv;
    std::map<int, std::deque<std::vector<char>>> inbox;  // source rank -> messages in arrival order
    std::map<int, int> out_fd;                           // peer -> connected socket
    bool closing = false;
};

struct Op {
    int kind;  // 0 send, 1 recv
    Comm* comm;
    const void* src;
    void* dst;
    size_t bytes;
    int peer;
    hipStream_t stream;
};

thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_deferred;

std::string path_of(const std::string& token, int rank) { return "/tmp/atx-rccl-stub-" + token + "-" + std::to_string(rank); }

bool read_all(int fd, void* p, size_t n) {
    char* c = static_cast<char*>(p);
    while (n) {
        const ssize_t got = ::read(fd, c, n);
        if (got <= 0) return false;
        c += got;
        n -= (size_t)got;
    }
    return true;
}

bool write_all(int fd, const void* p, size_t n) {
    const char* c = static_cast<const char*>(p);
    while (n) {
        const ssize_t put = ::write(fd, c, n);
        if (put <= 0) return false;
        c += put;
        n -= (size_t)put;
    }
    return true;
}

void reader_loop(Comm* c, int fd) {
    for (;;) {
        int32_t source;
        int64_t bytes;
        if (!read_all(fd, &source, sizeof(source)) || !read_all(fd, &bytes, sizeof(bytes))) break;
        std::vector<char> payload((size_t)bytes);
        if (bytes && !read_all(fd, payload.data(), (size_t)bytes)) break;
        {
            std::lock_guard<std::mutex> lock(c->mu);
            c->inbox[source].push_back(std::move(payload));
        }
        c->cv.notify_all();
    }
    ::close(fd);
}

void acceptor_loop(Comm* c) {
    for (;;) {
        const int fd = ::accept(c->listen_fd, nullptr, nullptr);
        if (fd < 0) return;  // listen socket closed
        std::lock_guard<std::mutex> lock(c->mu);
        if (c->closing) {
            ::close(fd);
            return;
        }
        c->readers.emplace_back(reader_loop, c, fd);
    }
}

int connect_to(Comm* c, int peer) {
    auto it = c->out_fd.find(peer);
    if (it != c->out_fd.end()) return it->second;
    const std::string path = path_of(c->token, peer);
    for (int attempt = 0; attempt < 6000; ++attempt) {  // the peer may not be listening yet: up to 60 s
        const int fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
        sockaddr_un addr{};
        addr.sun_family = AF_UNIX;
        std::strncpy(addr.sun_path, path.c_str(), sizeof(addr.sun_path) - 1);
        if (::connect(fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) == 0) {
            c->out_fd[peer] = fd;
            return fd;
        }
        ::close(fd);
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    return -1;
}

int run_send(const Op& op) {
    if (hipStreamSynchronize(op.stream) != hipSuccess) return 1;
    std::vector<char> host(op.bytes);
    if (op.bytes && hipMemcpy(host.data(), op.src, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const int fd = connect_to(op.comm, op.peer);
    if (fd < 0) return 2;
    const int32_t source = op.comm->rank;
    const int64_t bytes = (int64_t)op.bytes;
    if (!write_all(fd, &source, sizeof(source)) || !write_all(fd, &bytes, sizeof(bytes)) || !write_all(fd, host.data(), op.bytes)) return 2;
    return 0;
}

int run_recv(const Op& op) {
    std::vector<char> payload;
    {
        std::unique_lock<std::mutex> lock(op.comm->mu);
        if (!op.comm->cv.wait_for(lock, std::chrono::seconds(120), [&] { return !op.comm->inbox[op.peer].empty(); })) return 3;
        payload = std::move(op.comm->inbox[op.peer].front());
        op.comm->inbox[op.peer].pop_front();
    }
    if (payload.size() != op.bytes) {  // the very mismatch this stub exists to catch
        std::fprintf(stderr, "rccl_stub: rank %d expected %zu bytes from rank %d, got %zu\n", op.comm->rank, op.bytes, op.peer, payload.size());
        return 4;
    }
    if (hipStreamSynchronize(op.stream) != hipSuccess) return 1;
    if (op.bytes && hipMemcpy(op.dst, payload.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}

int submit(std::vector<Op> ops) {
    if (g_group_depth > 0) {
        for (auto& o : ops) g_deferred.push_back(o);
        return 0;
    }
    for (const auto& o : ops)
        if (o.kind == 0)
            if (int e = run_send(o)) return e;
    for (const auto& o : ops)
        if (o.kind == 1)
            if (int e = run_recv(o)) return e;
    return 0;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int* v) {
    *v = 29999;  // "2.99.99": recognisably not a real release
    return 0;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id->internal, 0, sizeof(id->internal));
    std::random_device rd;
    std::snprintf(id->internal, sizeof(id->internal), "%08x%08x", rd(), rd());
    return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
    Comm* c = new Comm;
    c->world = world;
    c->rank = rank;
    c->token = std::string(id.internal, strnlen(id.internal, 64));
    const std::string path = path_of(c->token, rank);
    ::unlink(path.c_str());
    c->listen_fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un addr{};
    addr.sun_family = AF_UNIX;
    std::strncpy(addr.sun_path, path.c_str(), sizeof(addr.sun_path) - 1);
    if (::bind(c->listen_fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) != 0 || ::listen(c->listen_fd, 64) != 0) {
        delete c;
        return 2;
    }
    c->acceptor = std::thread(acceptor_loop, c);
    *comm = reinterpret_cast<ncclComm_t>(c);
    return 0;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    {
        std::lock_guard<std::mutex> lock(c->mu);
        c->closing = true;
    }
    for (auto& kv : c->out_fd) ::close(kv.second);  // the peers' readers see end-of-file
    ::shutdown(c->listen_fd, SHUT_RDWR);
    ::close(c->listen_fd);
    if (c->acceptor.joinable()) c->acceptor.join();
    for (auto& t : c->readers)
        if (t.joinable()) t.detach();  // they end when the peers close their side
    ::unlink(path_of(c->token, c->rank).c_str());
    // (the Comm object is leaked on purpose: detached readers may still touch it)
    return 0;
}

const char* ncclGetErrorString(ncclResult_t e) {
    switch (e) {
        case 0: return "success";
        case 1: return "HIP error in the stub";
        case 2: return "socket error in the stub";
        case 3: return "timed out waiting for a message";
        case 4: return "message size differs from the posted receive";
        default: return "unknown stub error";
    }
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t /*bytes*/, int peer, ncclComm_t comm, hipStream_t stream) {
    return submit({Op{0, reinterpret_cast<Comm*>(comm), buf, nullptr, count, peer, stream}});
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t stream) {
    return submit({Op{1, reinterpret_cast<Comm*>(comm), nullptr, buf, count, peer, stream}});
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t, int root, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    std::vector<Op> ops;
    if (c->rank == root) {
        for (int p = 0; p < c->world; ++p)
            if (p != root) ops.push_back(Op{0, c, send, nullptr, count, p, stream});
        if (send != recv && count) {
            if (hipMemcpyAsync(recv, send, count, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
        }
    } else {
        ops.push_back(Op{1, c, nullptr, recv, count, root, stream});
    }
    return submit(std::move(ops));
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t comm, hipStream_t stream) {
    // every rank sends its block to every peer and receives every peer's block into that peer's slot
    Comm* c = reinterpret_cast<Comm*>(comm);
    char* out = static_cast<char*>(recv);
    std::vector<Op> ops;
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        ops.push_back(Op{0, c, send, nullptr, count, p, stream});
        ops.push_back(Op{1, c, nullptr, out + (size_t)p * count, count, p, stream});
    }
    char* own = out + (size_t)c->rank * count;
    if (send != own && count) {
        if (hipMemcpyAsync(own, send, count, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
    }
    return submit(std::move(ops));
}

ncclResult_t ncclGroupStart(void) {
    ++g_group_depth;
    return 0;
}

ncclResult_t ncclGroupEnd(void) {
    if (--g_group_depth > 0) return 0;
    std::vector<Op> ops;
    ops.swap(g_deferred);
    return submit(std::move(ops));
}

}  // extern "C"
