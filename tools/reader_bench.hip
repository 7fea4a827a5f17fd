// tools/reader_bench.hip -- what bounds the file readers of kg_scan.hip (pread of page-cache / tmpfs bytes into pinned memory, then H2D)?
//   hipcc --offload-arch=gfx950 -O2 tools/reader_bench.hip -o /tmp/reader_bench -lpthread && /tmp/reader_bench DIR [GB]
// Every method gets a FRESH file (written by 8 threads just before, as a run's inputs are: read exactly once) and reads it once, in
// 8 MiB segments dealt round-robin to T threads, each into its own pinned buffer: GB/s of the file.  The first read of a file's
// pages is what a real run pays (the second pass of a page-cache file is several times faster: pages already on the active list).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
constexpr size_t SEG = 8u << 20;
static size_t g_size;
static std::string g_path;

static void fresh_file() {
    unlink(g_path.c_str());
    const int fd = open(g_path.c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0600);
    if (ftruncate(fd, (off_t)g_size)) {}
    std::vector<std::thread> th;
    const size_t n_seg = g_size / SEG;
    std::atomic<size_t> next{0};
    for (int t = 0; t < 8; ++t) th.emplace_back([&, t] {
        std::vector<uint8_t> b(SEG, (uint8_t)('A' + t));
        for (;;) { const size_t i = next++; if (i >= n_seg) break; if (pwrite(fd, b.data(), SEG, (off_t)(i * SEG)) != (ssize_t)SEG) break; }
    });
    for (auto& x : th) x.join();
    close(fd);
}

template <class F>
static void run(const char* name, int T, int flags, F body /* (fd, thread, segment index, buffer) */, bool fresh = true, std::function<void(int)> prep = nullptr) {
    if (fresh) fresh_file();
    const int fd = open(g_path.c_str(), O_RDONLY | flags);
    if (fd < 0) { printf("%-58s open failed\n", name); return; }
    if (prep) prep(fd);
    std::vector<uint8_t*> bufs(T);
    for (int t = 0; t < T; ++t) if (hipHostMalloc((void**)&bufs[t], SEG, hipHostMallocNonCoherent) != hipSuccess) { printf("alloc failed\n"); return; }
    const size_t n_seg = g_size / SEG;
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    const double t0 = now();
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] { for (;;) { const size_t i = next++; if (i >= n_seg) break; if (!body(fd, t, i, bufs[t])) { ++bad; break; } } });
    for (auto& x : th) x.join();
    const double dt = now() - t0;
    printf("%-58s T=%2d  %7.2f GB/s%s\n", name, T, n_seg * SEG / dt / 1e9, bad ? "  (FAILED)" : "");
    fflush(stdout);
    for (int t = 0; t < T; ++t) hipHostFree(bufs[t]);
    close(fd);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s DIR [GB]\n", argv[0]); return 2; }
    g_path = std::string(argv[1]) + "/katgpu_reader_bench.bin";
    g_size = (size_t)((argc > 2 ? atof(argv[2]) : 4.0) * 1e9) / SEG * SEG;
    printf("files of %.2f GB in %s, %u hardware threads\n", g_size / 1e9, argv[1], std::thread::hardware_concurrency());
    auto pread_seg = [](int fd, int, size_t i, uint8_t* b) { size_t got = 0; while (got < SEG) { const ssize_t r = pread(fd, b + got, SEG - got, (off_t)(i * SEG + got)); if (r <= 0) return false; got += (size_t)r; } return true; };
    for (int T : {8, 16, 32, 64}) run("pread, first read of the file", T, 0, pread_seg);
    run("pread, second read (no fresh file)", 16, 0, pread_seg, false);
    run("pread after posix_fadvise(NOREUSE), first read", 16, 0, pread_seg, true, [](int fd) { posix_fadvise(fd, 0, 0, POSIX_FADV_NOREUSE); });
    run("pread after posix_fadvise(SEQUENTIAL), first read", 16, 0, pread_seg, true, [](int fd) { posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL); });
    run("pread O_DIRECT, first read", 16, O_DIRECT, pread_seg);
    {   // mmap + memcpy
        for (int populate = 0; populate < 2; ++populate) {
            fresh_file();
            const int fd = open(g_path.c_str(), O_RDONLY);
            uint8_t* m = (uint8_t*)mmap(nullptr, g_size, PROT_READ, MAP_SHARED, fd, 0);
            if (m == MAP_FAILED) { printf("mmap failed\n"); close(fd); continue; }
            auto body = [m, populate](int, int, size_t i, uint8_t* b) {
#ifdef MADV_POPULATE_READ
                if (populate) madvise(m + i * SEG, SEG, MADV_POPULATE_READ);
#endif
                memcpy(b, m + i * SEG, SEG); return true; };
            run(populate ? "mmap + MADV_POPULATE_READ per segment + memcpy, first read" : "mmap + memcpy, first read", 16, 0, body, false);
            munmap(m, g_size); close(fd);
        }
    }
    {   // as the product does it: pread + H2D (one copy in flight per thread), first read
        uint8_t* dev = nullptr;
        hipMalloc((void**)&dev, (size_t)64 * SEG);
        for (int T : {16, 32}) {
            std::vector<hipStream_t> st(T);
            for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            auto body = [&](int fd, int t, size_t i, uint8_t* b) {
                size_t got = 0; while (got < SEG) { const ssize_t r = pread(fd, b + got, SEG - got, (off_t)(i * SEG + got)); if (r <= 0) return false; got += (size_t)r; }
                return hipMemcpyAsync(dev + (i % 64) * SEG, b, SEG, hipMemcpyHostToDevice, st[t]) == hipSuccess && hipStreamSynchronize(st[t]) == hipSuccess; };
            run("pread + H2D, first read", T, 0, body);
            for (auto& s : st) hipStreamDestroy(s);
        }
    }
    unlink(g_path.c_str());
    return 0;
}
