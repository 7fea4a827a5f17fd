// tests/native/fake_rccl.cc -- TEST INFRASTRUCTURE: a stand-in for librccl that lets several ranks SHARING ONE GPU run the RCCL branch
// of kat_amd/csrc/kg_comm.hip (real RCCL refuses two ranks on one device, and the GPU boxes of this project have one).  It implements
// exactly the ten entry points kg_comm.hip resolves through dlopen -- ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy,
// ncclGroupStart / ncclGroupEnd, ncclSend / ncclRecv, ncclAllGather, ncclAllReduce, ncclGetErrorString -- with RCCL's matching rules
// (the n-th send to a peer meets the n-th receive from it; a group's operations happen together) over files in /dev/shm and
// hipMemcpy on the caller's stream.  Stronger than the real thing in one respect: a group has LANDED when ncclGroupEnd returns.
// Selected by KATGPU_RCCL_LIB under KATGPU_TESTING=1 (kg_comm.hip); never linked into, or shipped with, the product.
//   hipcc -shared -fPIC -O1 tests/native/fake_rccl.cc -o <dir>/libfakerccl.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Op { bool send; const void* src; void* dst; size_t bytes; int peer; hipStream_t stream; };

struct FakeComm {
    std::string token;
    int rank = 0, world = 1;
    std::vector<uint64_t> sent, received;          // per peer: messages so far
    uint64_t coll = 0;                             // collectives so far
    std::vector<std::string> mine;                 // files this rank created (removed at destroy)
};

thread_local int g_depth = 0;
thread_local std::vector<std::pair<FakeComm*, Op>> g_ops;

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    case ncclFloat16: return 2;
    default: return 0;
    }
}
std::string fname(const FakeComm* c, const char* what, int a, int b, uint64_t seq) {
    char buf[200];
    snprintf(buf, sizeof buf, "/dev/shm/fakerccl-%s-%s-%d-%d-%llu", c->token.c_str(), what, a, b, (unsigned long long)seq);
    return buf;
}
bool put_file(FakeComm* c, const std::string& name, const void* host, size_t n) {
    const std::string tmp = name + ".tmp";
    const int fd = ::open(tmp.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
    if (fd < 0) return false;
    size_t off = 0;
    while (off < n) { const ssize_t w = ::write(fd, (const char*)host + off, n - off); if (w <= 0) { ::close(fd); return false; } off += (size_t)w; }
    ::close(fd);
    if (rename(tmp.c_str(), name.c_str()) != 0) return false;          // published whole, or not at all
    c->mine.push_back(name);
    return true;
}
bool get_file(const std::string& name, void* host, size_t n) {
    const auto t0 = std::chrono::steady_clock::now();
    int fd;
    while ((fd = ::open(name.c_str(), O_RDONLY)) < 0) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    size_t off = 0;
    while (off < n) { const ssize_t g = ::read(fd, (char*)host + off, n - off); if (g <= 0) { ::close(fd); return false; } off += (size_t)g; }
    ::close(fd);
    return true;
}

ncclResult_t run_group() {
    // what the operations read must have been produced: everything queued so far on their streams
    for (auto& e : g_ops) if (hipStreamSynchronize(e.second.stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<char> host;
    for (auto& e : g_ops) {                                            // sends first: a file in tmpfs never blocks its writer
        FakeComm* c = e.first; const Op& o = e.second;
        if (!o.send) continue;
        host.resize(o.bytes);
        if (o.bytes && hipMemcpy(host.data(), o.src, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (!put_file(c, fname(c, "p2p", c->rank, o.peer, c->sent[(size_t)o.peer]++), host.data(), o.bytes)) return ncclSystemError;
    }
    for (auto& e : g_ops) {
        FakeComm* c = e.first; const Op& o = e.second;
        if (o.send) continue;
        host.resize(o.bytes);
        if (!get_file(fname(c, "p2p", o.peer, c->rank, c->received[(size_t)o.peer]++), host.data(), o.bytes)) return ncclSystemError;
        if (o.bytes && (hipMemcpyAsync(o.dst, host.data(), o.bytes, hipMemcpyHostToDevice, o.stream) != hipSuccess || hipStreamSynchronize(o.stream) != hipSuccess)) return ncclUnhandledCudaError;
    }
    g_ops.clear();
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    unsigned char rnd[8] = {0};
    FILE* f = fopen("/dev/urandom", "rb");
    if (f) { if (fread(rnd, 1, sizeof rnd, f) != sizeof rnd) rnd[0] = (unsigned char)getpid(); fclose(f); }
    snprintf(id->internal, sizeof id->internal, "fake%02x%02x%02x%02x%02x%02x%02x%02x", rnd[0], rnd[1], rnd[2], rnd[3], rnd[4], rnd[5], rnd[6], rnd[7]);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || strncmp(id.internal, "fake", 4) != 0) return ncclInvalidArgument;
    if (getenv("FAKE_RCCL_HANG_INIT")) for (;;) sleep(1);                // a bootstrap that never comes back (test_gpu_comm.py: the init deadline)
    FakeComm* c = new FakeComm();
    c->token.assign(id.internal, strnlen(id.internal, 24));
    c->rank = rank; c->world = nranks;
    c->sent.assign((size_t)nranks, 0); c->received.assign((size_t)nranks, 0);
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    if (!c) return ncclInvalidArgument;
    for (auto& n : c->mine) ::unlink(n.c_str());
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth) return ncclSuccess;
    return run_group();
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    if (!c || peer < 0 || peer >= c->world || !type_bytes(t)) return ncclInvalidArgument;
    g_ops.push_back({c, Op{true, buf, nullptr, count * type_bytes(t), peer, stream}});
    return g_depth ? ncclSuccess : run_group();
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    if (!c || peer < 0 || peer >= c->world || !type_bytes(t)) return ncclInvalidArgument;
    g_ops.push_back({c, Op{false, nullptr, buf, count * type_bytes(t), peer, stream}});
    return g_depth ? ncclSuccess : run_group();
}

// every rank's `count` elements, in rank order, into every rank's recvbuff
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t stream) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    const size_t n = count * type_bytes(t);
    if (!c || !type_bytes(t)) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<char> host(n);
    if (n && hipMemcpy(host.data(), sendbuff, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    const uint64_t seq = c->coll++;
    if (!put_file(c, fname(c, "coll", c->rank, 0, seq), host.data(), n)) return ncclSystemError;
    for (int r = 0; r < c->world; ++r) {
        if (!get_file(fname(c, "coll", r, 0, seq), host.data(), n)) return ncclSystemError;
        if (n && hipMemcpyAsync((char*)recvbuff + (size_t)r * n, host.data(), n, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    if (!c || t != ncclUint64 || op != ncclSum) return ncclInvalidArgument;        // (all that kg_comm.hip asks for)
    const size_t n = count * 8;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<uint64_t> mine(count), other(count), sum(count, 0);
    if (n && hipMemcpy(mine.data(), sendbuff, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    const uint64_t seq = c->coll++;
    if (!put_file(c, fname(c, "coll", c->rank, 0, seq), mine.data(), n)) return ncclSystemError;
    for (int r = 0; r < c->world; ++r) {
        if (!get_file(fname(c, "coll", r, 0, seq), other.data(), n)) return ncclSystemError;
        for (size_t i = 0; i < count; ++i) sum[i] += other[i];
    }
    if (n && (hipMemcpyAsync(recvbuff, sum.data(), n, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled hip error (fake rccl)";
    case ncclSystemError: return "system error (fake rccl: /dev/shm)";
    case ncclInvalidArgument: return "invalid argument (fake rccl)";
    case ncclInvalidUsage: return "invalid usage (fake rccl)";
    default: return "error (fake rccl)";
    }
}

}  // extern "C"
