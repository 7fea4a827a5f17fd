// Diagnostic (host only, no GPU): what the ingest front ends deliver into a sink that does nothing.
//   g++ -O2 -std=c++17 -I include tools/ingest_host_bench.cc kat_amd/csrc/kg_ingest.cpp -o /tmp/ingest_host_bench -lz -lpthread
//   /tmp/ingest_host_bench FILE [threads...]
#include "../kat_amd/csrc/kg_ingest.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s FILE [threads...]\n", argv[0]); return 2; }
    const char* path = argv[1];
    const double gb = kg::file_size_or_zero(path) / 1e9;
    printf("host threads: %u, file %.2f GB\n", std::thread::hardware_concurrency(), gb);
    auto run = [&](const char* label, bool team) {
        size_t bytes = 0; uint64_t sum = 0;
        auto sink = [&](const uint8_t* p, size_t n) { bytes += n; sum += p[0] + p[n - 1]; return 0; };
        std::string err;
        auto t0 = std::chrono::steady_clock::now();
        int rc;
        if (team) rc = kg::parse_file_parallel(path, 0, sink, &err);
        else {
            kg::SeqFileParser ps;
            rc = ps.open(path, 0, &err);
            while (!rc) { const uint8_t* p; size_t n; rc = ps.next(&p, &n, &err); if (rc || !n) break; sink(p, n); }
        }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%-22s rc=%d  %.3f s  %.2f GB/s of file  (%zu base bytes, check %llu)\n", label, rc, dt, gb / dt, bytes, (unsigned long long)sum);
    };
    run("stream", false);
    setenv("KATGPU_INGEST_MIN_BYTES", "0", 1);
    for (int i = 2; i < argc; ++i) {
        setenv("KATGPU_INGEST_THREADS", argv[i], 1);
        char label[64]; snprintf(label, sizeof label, "team %s threads", argv[i]);
        run(label, true);
        run(label, true);
    }
    return 0;
}
