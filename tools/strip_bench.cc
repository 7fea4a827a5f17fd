// What one reader thread of the large-FASTQ ingest does per record (kg_ingest.cpp: strip_fastq_records) on 1 GiB of synthetic 150-base records in memory.
//   hipcc -O3 -std=c++17 -fPIC -c kat_amd/csrc/kg_ingest.cpp -I include -o /tmp/kg_ingest.o && g++ -O2 -std=c++17 tools/strip_bench.cc /tmp/kg_ingest.o -lz -lpthread -o /tmp/strip_bench && /tmp/strip_bench
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <vector>
#include <string>
namespace kg { bool strip_fastq_records(const uint8_t* p, size_t n, uint8_t* out, size_t* out_n); }
int main() {
    std::string rec;
    std::vector<uint8_t> data;
    const size_t target = (size_t)1 << 30;
    uint64_t x = 88172645463325252ULL;
    size_t i = 0;
    while (data.size() < target) {
        char hdr[64]; int hl = snprintf(hdr, sizeof hdr, "@lib1.%zu/1\n", i++);
        data.insert(data.end(), hdr, hdr + hl);
        for (int j = 0; j < 150; ++j) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; data.push_back("ACGT"[x & 3]); }
        data.push_back('\n'); data.push_back('+'); data.push_back('\n');
        for (int j = 0; j < 150; ++j) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; data.push_back((uint8_t)(33 + (x % 41))); }
        data.push_back('\n');
    }
    std::vector<uint8_t> out(data.size() / 2 + 1024);
    for (int rep = 0; rep < 4; ++rep) {
        size_t on = 0;
        auto t0 = std::chrono::steady_clock::now();
        bool ok = kg::strip_fastq_records(data.data(), data.size(), out.data(), &on);
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("ok=%d  %.2f GB/s in, %zu records, %.1f ns/record, out %zu\n", ok, data.size() / s / 1e9, i, s / i * 1e9, on);
    }
}
