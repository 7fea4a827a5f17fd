// kg_jf.cpp -- Jellyfish "binary/sorted" hash files (.jf): reader and writer.  Pure host code.
//
// Replaces JellyfishHelper::dumpHash / HashLoader::loadHash (lib/src/jellyfish_helper.cc:248-256,97-187) and the parts
// of Jellyfish they stand on: generic_file_header::write/read (JF/include/jellyfish/generic_file_header.hpp:96-153),
// file_header (file_header.hpp:34-108), binary_writer / binary_reader (binary_dumper.hpp:47-51,114-119) and the
// order sorted_dumper emits (sorted_dumper.hpp:80-112 with mer_heap.hpp:33-37).
//
// File layout: 9 decimal digits = length L of the JSON header incl. zero padding; the JSON (keys in alphabetical
// order, as jsoncpp's FastWriter emits them) padded so that 9 + L is a multiple of `alignment` (8); then fixed-width
// records: ceil(key_len/8) key bytes (the 2-bit packed k-mer, little endian) + counter_len count bytes (little endian,
// saturated at 2^(8*counter_len)-1, KAT hard-codes counter_len 4: lib/src/input_handler.cc:196), sorted by
// (M * kmer) & (size-1) and then by k-mer, where M is the r x key_len GF(2) matrix stored in the header as "matrix1"
// (bit i of the k-mer selects column key_len-1-i: rectangular_binary_matrix.hpp:206-240).  Checked against the
// reference's own fixture tests/data/ecoli.header.jf27 (tests/test_jf.py).
#include "../../include/katgpu.h"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <unistd.h>
#include <vector>

namespace {

thread_local std::string g_jf_err;

// ---- minimal JSON field access for the flat header object ----
bool json_find(const std::string& js, const char* key, size_t* val_pos) {
    const std::string pat = std::string("\"") + key + "\":";
    size_t p = js.find(pat);
    if (p == std::string::npos) return false;
    *val_pos = p + pat.size();
    return true;
}
bool json_uint(const std::string& js, const char* key, uint64_t* out) {
    size_t p;
    if (!json_find(js, key, &p)) return false;
    while (p < js.size() && isspace((unsigned char)js[p])) ++p;
    if (p >= js.size() || !isdigit((unsigned char)js[p])) return false;
    *out = strtoull(js.c_str() + p, nullptr, 10);
    return true;
}
bool json_string(const std::string& js, const char* key, std::string* out) {
    size_t p;
    if (!json_find(js, key, &p)) return false;
    while (p < js.size() && isspace((unsigned char)js[p])) ++p;
    if (p >= js.size() || js[p] != '"') return false;
    std::string v;                                               // up to the closing quote; escapes (jsoncpp writes \" \\ \n ... \uXXXX) undone
    for (size_t i = p + 1; i < js.size(); ++i) {
        const char ch = js[i];
        if (ch == '"') { *out = v; return true; }
        if (ch != '\\' || i + 1 >= js.size()) { v += ch; continue; }
        const char e = js[++i];
        switch (e) {
            case 'n': v += '\n'; break; case 't': v += '\t'; break; case 'r': v += '\r'; break;
            case 'b': v += '\b'; break; case 'f': v += '\f'; break;
            case 'u': if (i + 4 < js.size()) { v += (char)strtoul(js.substr(i + 1, 4).c_str(), nullptr, 16); i += 4; } break;
            default: v += e;                                     // \" \\ \/
        }
    }
    return false;
}
// a string value as Json::FastWriter emits it (the reference writes its header through jsoncpp, which escapes)
std::string json_quote(const std::string& v) {
    std::string o = "\"";
    for (const unsigned char ch : v) {
        switch (ch) {
            case '"': o += "\\\""; break; case '\\': o += "\\\\"; break; case '\n': o += "\\n"; break; case '\t': o += "\\t"; break;
            case '\r': o += "\\r"; break; case '\b': o += "\\b"; break; case '\f': o += "\\f"; break;
            default:
                if (ch < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04X", ch); o += b; } else o += (char)ch;
        }
    }
    return o + "\"";
}
bool json_bool(const std::string& js, const char* key, bool def) {
    size_t p;
    if (!json_find(js, key, &p)) return def;
    while (p < js.size() && isspace((unsigned char)js[p])) ++p;
    return js.compare(p, 4, "true") == 0;
}

uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// r x c GF(2) matrix as c columns of r bits; the last r columns (those multiplying the low r bits of the k-mer) form an
// invertible square block, the property Jellyfish's randomize_pseudo_inverse guarantees (lib/rectangular_binary_matrix.cc:209-216)
std::vector<uint64_t> random_matrix(unsigned r, unsigned c, uint64_t seed) {
    const uint64_t mask = r >= 64 ? ~0ULL : ((1ULL << r) - 1);
    std::vector<uint64_t> cols(c);
    for (;;) {
        for (auto& x : cols) x = splitmix(seed) & mask;
        if (c < r) return cols;
        std::vector<uint64_t> m(cols.end() - r, cols.end());        // Gaussian elimination on the square block
        unsigned rank = 0;
        for (unsigned bit = 0; bit < r; ++bit) {
            unsigned piv = rank;
            while (piv < r && !(m[piv] >> bit & 1)) ++piv;
            if (piv == r) break;
            std::swap(m[rank], m[piv]);
            for (unsigned j = 0; j < r; ++j) if (j != rank && (m[j] >> bit & 1)) m[j] ^= m[rank];
            ++rank;
        }
        if (rank == r) return cols;
    }
}

typedef unsigned __int128 u128;                               // a k-mer of up to 64 bases (k > 32: "wide", include/katgpu.h)

inline uint64_t matrix_times(const std::vector<uint64_t>& cols, u128 key) {
    const unsigned c = (unsigned)cols.size();
    uint64_t res = 0;
    for (unsigned i = 0; i < c && key; ++i, key >>= 1) if (key & 1) res ^= cols[c - 1 - i];
    return res;
}

struct Rec { uint64_t pos; u128 key; uint64_t count; };

}  // namespace

extern "C" const char* katgpu_jf_last_error(void) { return g_jf_err.c_str(); }

// keys_hi == nullptr: one-word k-mers
static int write_records(const char* path, uint32_t k, int canonical, const uint64_t* keys_hi, const uint64_t* keys, const uint64_t* counts, size_t n) {
    const unsigned key_len = 2 * k, key_bytes = (key_len + 7) / 8, counter_len = 4;
    // size: the power of two Jellyfish would have needed for n entries (HashLoader sizes 2n rounded up, jellyfish_helper.cc:144-145)
    unsigned r = 1;
    while (((uint64_t)1 << r) < std::max<uint64_t>(2 * (uint64_t)n, 2)) ++r;
    if (r > key_len) r = key_len;
    const uint64_t size = (uint64_t)1 << r;
    const std::vector<uint64_t> cols = random_matrix(r, key_len, 0x6B61746770750000ULL ^ ((uint64_t)k << 8) ^ (uint64_t)n);
    std::vector<Rec> recs(n);
    for (size_t i = 0; i < n; ++i) {
        const u128 key = ((u128)(keys_hi ? keys_hi[i] : 0) << 64) | keys[i];
        recs[i] = {matrix_times(cols, key) & (size - 1), key, counts[i]};
    }
    std::sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) { return a.pos != b.pos ? a.pos < b.pos : a.key < b.key; });

    // ---- header (alphabetical keys, terse, like jsoncpp's FastWriter) ----
    char host[256] = "localhost", cwd[4096] = ".", when[64] = "";
    gethostname(host, sizeof host - 1);
    if (!getcwd(cwd, sizeof cwd)) strcpy(cwd, ".");
    time_t now = time(nullptr);
    strftime(when, sizeof when, "%a %b %e %H:%M:%S %Y", localtime(&now));
    std::string js = "{\"alignment\":8,\"canonical\":";
    js += canonical ? "true" : "false";
    js += ",\"cmdline\":[\"katgpu\"],\"counter_len\":" + std::to_string(counter_len);
    js += ",\"exe_path\":\"katgpu\",\"format\":\"binary/sorted\",\"hostname\":" + json_quote(host);
    js += ",\"key_len\":" + std::to_string(key_len) + ",\"matrix1\":{\"c\":" + std::to_string(key_len) + ",\"columns\":[";
    for (unsigned i = 0; i < key_len; ++i) { if (i) js += ','; js += std::to_string(cols[i]); }
    js += "],\"r\":" + std::to_string(r) + "},\"max_reprobe\":126,\"pwd\":" + json_quote(cwd) + ",\"reprobes\":[1";
    for (unsigned i = 1; i <= 126; ++i) js += "," + std::to_string((uint64_t)i * (i + 1) / 2);      // JF/lib/storage.cc:20-50
    js += "],\"size\":" + std::to_string(size) + ",\"time\":\"" + when + "\",\"val_len\":7}";
    size_t hlen = js.size();
    const size_t rem = (9 + js.size()) % 8;
    if (rem) hlen += 8 - rem;

    FILE* f = fopen(path, "wb");
    if (!f) { g_jf_err = std::string("cannot open ") + path + " for writing"; return KATGPU_ERR_IO; }
    fprintf(f, "%09zu", hlen);
    fwrite(js.data(), 1, js.size(), f);
    for (size_t i = js.size(); i < hlen; ++i) fputc('\0', f);
    std::vector<uint8_t> buf;
    buf.reserve((size_t)(key_bytes + counter_len) * std::min<size_t>(n, 1 << 20));
    for (size_t i = 0; i < n; ++i) {
        const uint64_t v = std::min<uint64_t>(recs[i].count, 0xFFFFFFFFULL);                        // binary_writer::write saturates
        for (unsigned b = 0; b < key_bytes; ++b) buf.push_back((uint8_t)(recs[i].key >> (8 * b)));
        for (unsigned b = 0; b < counter_len; ++b) buf.push_back((uint8_t)(v >> (8 * b)));
        if (buf.size() >= ((size_t)8 << 20) || i + 1 == n) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
    }
    const bool ok = fclose(f) == 0;
    if (!ok) { g_jf_err = std::string("write error on ") + path; return KATGPU_ERR_IO; }
    return KATGPU_OK;
}

extern "C" int katgpu_jf_write_records(const char* path, uint32_t k, int canonical, const uint64_t* keys, const uint64_t* counts, size_t n) {
    if (!path || k < 1 || k > 32 || (n && (!keys || !counts))) return KATGPU_ERR_INVALID_ARG;
    return write_records(path, k, canonical, nullptr, keys, counts, n);
}

extern "C" int katgpu_jf_write_records_wide(const char* path, uint32_t k, int canonical, const uint64_t* keys_hi, const uint64_t* keys_lo,
                                            const uint64_t* counts, size_t n) {
    if (!path || k < 1 || k > KATGPU_MAX_K || (n && (!keys_hi || !keys_lo || !counts))) return KATGPU_ERR_INVALID_ARG;
    return write_records(path, k, canonical, keys_hi, keys_lo, counts, n);
}

// max_key_len: 64 for the one-word form (keys_hi == nullptr), 2 * KATGPU_MAX_K for the wide one
static int read_records(const char* path, unsigned max_key_len, uint32_t* k, int* canonical, uint64_t** keys_hi, uint64_t** keys, uint64_t** counts, size_t* n) {
    *keys = *counts = nullptr; *n = 0;
    if (keys_hi) *keys_hi = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) { g_jf_err = std::string("Could not find input file at: ") + path + "; please check the path and try again."; return KATGPU_ERR_IO; }
    char digits[10] = {0};
    size_t got = fread(digits, 1, 9, f);
    bool ok = got == 9;
    for (int i = 0; ok && i < 9; ++i) ok = isdigit((unsigned char)digits[i]);
    const unsigned long hlen = ok ? strtoul(digits, nullptr, 10) : 0;
    std::string js(hlen, '\0');
    if (!ok || hlen < 2 || fread(&js[0], 1, hlen, f) != hlen || js[0] != '{') {
        fclose(f);
        g_jf_err = std::string("Failed to parse header of file: ") + path;                          // jellyfish_helper.cc:102-105
        return KATGPU_ERR_FORMAT;
    }
    while (!js.empty() && js.back() == '\0') js.pop_back();
    std::string format;
    uint64_t key_len = 0, counter_len = 0;
    json_string(js, "format", &format);
    if (format != "binary/sorted") {
        fclose(f);
        g_jf_err = format == "bloomcounter" ? "KAT does not currently support bloom counted kmer hashes.  Please create a binary hash with jellyfish or KAT and use that instead."
                 : format == "text/sorted" ? "Processing a text format hash will be painfully slow, so we don't support it.  Please create a binary hash with jellyfish or KAT and use that instead."
                 : "Unknown format '" + format + "'";                                               // jellyfish_helper.cc:111-119,181-185
        return KATGPU_ERR_FORMAT;
    }
    if (!json_uint(js, "key_len", &key_len) || !json_uint(js, "counter_len", &counter_len) || key_len == 0 || key_len % 2 || counter_len == 0 || counter_len > 8) {
        fclose(f);
        g_jf_err = std::string("Failed to parse header of file: ") + path;
        return KATGPU_ERR_FORMAT;
    }
    if (key_len > max_key_len) {
        fclose(f);
        g_jf_err = "k = " + std::to_string(key_len / 2) + " unsupported: " + (max_key_len == 64 ? "this entry point reads one-word k-mers (k <= 32)" : "this build keeps a k-mer in at most two 63-bit words (k <= " + std::to_string(KATGPU_MAX_K) + ")");
        return KATGPU_ERR_K;
    }
    const size_t key_bytes = (key_len + 7) / 8, rec = key_bytes + counter_len, offset = 9 + hlen;
    fseek(f, 0, SEEK_END);
    const size_t data_bytes = (size_t)ftell(f) - offset;
    if (data_bytes % rec != 0) {
        fclose(f);
        g_jf_err = "Size of database (" + std::to_string(data_bytes) + ") must be a multiple of the length of a record (" + std::to_string(rec) + ")";   // :162-167
        return KATGPU_ERR_FORMAT;
    }
    const size_t nrec = data_bytes / rec;
    uint64_t* kk = (uint64_t*)malloc(std::max<size_t>(nrec, 1) * 8);
    uint64_t* cc = (uint64_t*)malloc(std::max<size_t>(nrec, 1) * 8);
    uint64_t* kh = keys_hi ? (uint64_t*)malloc(std::max<size_t>(nrec, 1) * 8) : nullptr;
    if (!kk || !cc || (keys_hi && !kh)) { fclose(f); free(kk); free(cc); free(kh); return KATGPU_ERR_NOMEM; }
    fseek(f, (long)offset, SEEK_SET);
    std::vector<uint8_t> buf(rec * (1 << 16));
    size_t i = 0;
    while (i < nrec) {
        const size_t take = std::min<size_t>(nrec - i, 1 << 16);
        if (fread(buf.data(), rec, take, f) != take) { fclose(f); free(kk); free(cc); free(kh); g_jf_err = std::string("read error on ") + path; return KATGPU_ERR_IO; }
        for (size_t j = 0; j < take; ++j) {
            const uint8_t* p = buf.data() + j * rec;
            u128 key = 0; uint64_t cnt = 0;
            for (size_t b = 0; b < key_bytes; ++b) key |= (u128)p[b] << (8 * b);
            for (size_t b = 0; b < counter_len; ++b) cnt |= (uint64_t)p[key_bytes + b] << (8 * b);
            kk[i + j] = (uint64_t)key; cc[i + j] = cnt;
            if (kh) kh[i + j] = (uint64_t)(key >> 64);
        }
        i += take;
    }
    fclose(f);
    *k = (uint32_t)(key_len / 2);
    if (canonical) *canonical = json_bool(js, "canonical", false) ? 1 : 0;
    *keys = kk; *counts = cc; *n = nrec;
    if (keys_hi) *keys_hi = kh;
    return KATGPU_OK;
}

extern "C" int katgpu_jf_read_records(const char* path, uint32_t* k, int* canonical, uint64_t** keys, uint64_t** counts, size_t* n) {
    if (!path || !k || !keys || !counts || !n) return KATGPU_ERR_INVALID_ARG;
    return read_records(path, 64, k, canonical, nullptr, keys, counts, n);
}

extern "C" int katgpu_jf_read_records_wide(const char* path, uint32_t* k, int* canonical, uint64_t** keys_hi, uint64_t** keys_lo, uint64_t** counts, size_t* n) {
    if (!path || !k || !keys_hi || !keys_lo || !counts || !n) return KATGPU_ERR_INVALID_ARG;
    return read_records(path, 2 * KATGPU_MAX_K, k, canonical, keys_hi, keys_lo, counts, n);
}

// ---- device-level wrappers: InputHandler::loadHash / dump ----

extern "C" int katgpu_jf_load(katgpu_ctx* ctx, const char* path, katgpu_table** out) {
    if (!ctx || !out) return KATGPU_ERR_INVALID_ARG;
    *out = nullptr;
    uint32_t k = 0; int canonical = 0; uint64_t *keys_hi = nullptr, *keys = nullptr, *counts = nullptr; size_t n = 0;
    int rc = katgpu_jf_read_records_wide(path, &k, &canonical, &keys_hi, &keys, &counts, &n);
    if (rc) return rc;
    katgpu_table* t = nullptr;
    rc = katgpu_table_create(ctx, k, canonical, std::max<uint64_t>((uint64_t)(n / 0.6) + 1024, 1 << 16), 0, &t);
    if (!rc) rc = k > 32 ? katgpu_table_merge_host_wide(t, keys_hi, keys, counts, n)
                         : katgpu_table_merge_host(t, keys, counts, n);       // hash->add(reader.key(), reader.val()) per record (:172-174)
    free(keys_hi); free(keys); free(counts);
    if (rc) { if (t) katgpu_table_free(t); g_jf_err = katgpu_last_error(ctx); return rc; }
    *out = t;
    return KATGPU_OK;
}

extern "C" int katgpu_jf_dump(katgpu_table* t, const char* path) {
    if (!t || !path) return KATGPU_ERR_INVALID_ARG;
    size_t n = 0;
    if (katgpu_table_k(t) > 32) {
        int rc = katgpu_table_export_wide(t, nullptr, nullptr, nullptr, 0, &n);
        if (rc) return rc;
        std::vector<uint64_t> hi(std::max<size_t>(n, 1)), lo(std::max<size_t>(n, 1)), counts(std::max<size_t>(n, 1));
        if (n) { rc = katgpu_table_export_wide(t, hi.data(), lo.data(), counts.data(), n, &n); if (rc) return rc; }
        return katgpu_jf_write_records_wide(path, katgpu_table_k(t), katgpu_table_canonical(t), hi.data(), lo.data(), counts.data(), n);
    }
    int rc = katgpu_table_export(t, nullptr, nullptr, 0, &n);
    if (rc) return rc;
    std::vector<uint64_t> keys(std::max<size_t>(n, 1)), counts(std::max<size_t>(n, 1));
    if (n) { rc = katgpu_table_export(t, keys.data(), counts.data(), n, &n); if (rc) return rc; }
    return katgpu_jf_write_records(path, katgpu_table_k(t), katgpu_table_canonical(t), keys.data(), counts.data(), n);
}
