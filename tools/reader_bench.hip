// tools/reader_bench.hip -- what bounds the file readers of kg_scan.hip (pread of page-cache / tmpfs bytes into pinned memory, then H2D)?
//   hipcc --offload-arch=gfx950 -O2 tools/reader_bench.hip -o /tmp/reader_bench -lpthread && /tmp/reader_bench FILE [threads...]
// For each thread count and each kind of destination buffer -- malloc, hipHostMalloc (default / non-coherent / write-combined),
// malloc + hipHostRegister -- T threads pread the file in 8 MiB segments (round-robin) into their own buffer; GB/s of the file.
// Then the same through mmap + memcpy, and pread + H2D copy as the product does it.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
constexpr size_t SEG = 8u << 20;

enum Kind { MALLOC, PIN_DEFAULT, PIN_NONCOH, PIN_WC, REGISTERED, N_KINDS };
static const char* kind_name[] = {"malloc", "hipHostMalloc default", "hipHostMalloc non-coherent", "hipHostMalloc write-combined", "malloc + hipHostRegister"};

static uint8_t* get_buf(Kind k) {
    uint8_t* p = nullptr;
    if (k == MALLOC) { p = (uint8_t*)aligned_alloc(4096, SEG); memset(p, 1, SEG); }
    else if (k == PIN_DEFAULT) { if (hipHostMalloc((void**)&p, SEG, hipHostMallocDefault) != hipSuccess) p = nullptr; }
    else if (k == PIN_NONCOH) { if (hipHostMalloc((void**)&p, SEG, hipHostMallocNonCoherent) != hipSuccess) p = nullptr; }
    else if (k == PIN_WC) { if (hipHostMalloc((void**)&p, SEG, hipHostMallocWriteCombined) != hipSuccess) p = nullptr; }
    else { p = (uint8_t*)aligned_alloc(4096, SEG); memset(p, 1, SEG); if (hipHostRegister(p, SEG, hipHostRegisterDefault) != hipSuccess) { free(p); p = nullptr; } }
    return p;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s FILE [threads...]\n", argv[0]); return 2; }
    const int fd = open(argv[1], O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st)) { perror("open"); return 1; }
    const size_t size = (size_t)st.st_size, n_seg = size / SEG;
    std::vector<int> Ts;
    for (int i = 2; i < argc; ++i) Ts.push_back(atoi(argv[i]));
    if (Ts.empty()) Ts = {16, 32, 64};
    printf("file %.2f GB, %u hardware threads\n", size / 1e9, std::thread::hardware_concurrency());
    uint8_t* dev = nullptr;
    hipMalloc((void**)&dev, (size_t)64 * SEG);
    for (int T : Ts) {
        for (int k = 0; k < N_KINDS; ++k) {
            std::vector<uint8_t*> bufs(T);
            bool ok = true;
            for (int t = 0; t < T; ++t) { bufs[t] = get_buf((Kind)k); ok = ok && bufs[t]; }
            if (!ok) { printf("T=%2d %-32s allocation failed\n", T, kind_name[k]); continue; }
            for (int h2d = 0; h2d < (k == MALLOC ? 1 : 2); ++h2d) {
                std::atomic<size_t> next{0};
                std::vector<std::thread> th;
                const double t0 = now();
                for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                    hipStream_t s = nullptr;
                    if (h2d) { hipSetDevice(0); hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
                    for (;;) {
                        const size_t i = next++;
                        if (i >= n_seg) break;
                        size_t got = 0;
                        while (got < SEG) { const ssize_t r = pread(fd, bufs[t] + got, SEG - got, (off_t)(i * SEG + got)); if (r <= 0) break; got += (size_t)r; }
                        if (h2d) { hipMemcpyAsync(dev + (i % 64) * SEG, bufs[t], SEG, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }
                    }
                    if (s) hipStreamDestroy(s);
                });
                for (auto& x : th) x.join();
                const double dt = now() - t0;
                printf("T=%2d %-32s pread%s  %6.2f GB/s\n", T, kind_name[k], h2d ? " + H2D" : "      ", n_seg * SEG / dt / 1e9);
            }
            for (int t = 0; t < T; ++t) {
                if (k == MALLOC) free(bufs[t]);
                else if (k == REGISTERED) { hipHostUnregister(bufs[t]); free(bufs[t]); }
                else hipHostFree(bufs[t]);
            }
        }
        {   // mmap + memcpy into malloc'd buffers
            uint8_t* m = (uint8_t*)mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0);
            if (m != MAP_FAILED) {
                std::vector<uint8_t*> bufs(T);
                for (int t = 0; t < T; ++t) { hipHostMalloc((void**)&bufs[t], SEG, hipHostMallocDefault); }
                std::atomic<size_t> next{0};
                std::vector<std::thread> th;
                const double t0 = now();
                for (int t = 0; t < T; ++t) th.emplace_back([&, t] { for (;;) { const size_t i = next++; if (i >= n_seg) break; memcpy(bufs[t], m + i * SEG, SEG); } });
                for (auto& x : th) x.join();
                printf("T=%2d %-32s memcpy        %6.2f GB/s\n", T, "mmap -> hipHostMalloc default", n_seg * SEG / (now() - t0) / 1e9);
                for (int t = 0; t < T; ++t) hipHostFree(bufs[t]);
                munmap(m, size);
            }
        }
    }
    return 0;
}
