// matrix_io.cc -- metadata keys, dense matrix printing, CompCounters report, distance metrics, option parsing.
#include "kat_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <sstream>

namespace kat {

namespace mme {   // lib/include/kat/matrix_metadata_extractor.hpp:28-39
const char* const KEY_NB_COLUMNS = "# Columns:";
const char* const KEY_NB_ROWS = "# Rows:";
const char* const KEY_X_LABEL = "# XLabel:";
const char* const KEY_Y_LABEL = "# YLabel:";
const char* const KEY_Z_LABEL = "# ZLabel:";
const char* const KEY_INPUT_1 = "# Input 1:";
const char* const KEY_INPUT_2 = "# Input 2:";
const char* const KEY_KMER = "# Kmer value:";
const char* const KEY_TITLE = "# Title:";
const char* const KEY_MAX_VAL = "# MaxVal:";
const char* const KEY_TRANSPOSE = "# Transpose:";
const char* const MX_META_END = "###";
}

uint64_t Matrix64::getMaxVal() const {
    uint64_t mx = 0;
    for (uint64_t x : v) mx = std::max(mx, x);
    return mx;
}

void Matrix64::printMatrix(std::ostream& out) const {      // lib/include/kat/sparse_matrix.hpp:269-277
    std::string line;
    for (uint32_t i = 0; i < m; i++) {
        line.clear();
        line += std::to_string(get(i, 0));
        for (uint32_t j = 1; j < n; j++) { line += ' '; line += std::to_string(get(i, j)); }
        out << line << std::endl;
    }
}

// ---- CompCounters (lib/src/comp_counters.cc) ----
CompCounters::CompCounters(const std::string& p1, const std::string& p2, const std::string& p3, size_t dm_size)
    : spectrum1(dm_size, 0), spectrum2(dm_size, 0), shared_spectrum1(dm_size, 0), shared_spectrum2(dm_size, 0),
      hash1_path(p1), hash2_path(p2), hash3_path(p3) {}

void CompCounters::loadDevice(const uint64_t c[13], const uint64_t* sp) {
    hash1_total = c[0]; hash2_total = c[1]; hash3_total = c[2];
    hash1_distinct = c[3]; hash2_distinct = c[4]; hash3_distinct = c[5];
    hash1_only_total = c[6]; hash2_only_total = c[7]; hash1_only_distinct = c[8]; hash2_only_distinct = c[9];
    shared_hash1_total = c[10]; shared_hash2_total = c[11]; shared_distinct = c[12];
    const size_t n = spectrum1.size();
    std::copy(sp, sp + n, spectrum1.begin());
    std::copy(sp + n, sp + 2 * n, spectrum2.begin());
    std::copy(sp + 2 * n, sp + 3 * n, shared_spectrum1.begin());
    std::copy(sp + 3 * n, sp + 4 * n, shared_spectrum2.begin());
}

// boost::filesystem::path's operator<< writes the path double-quoted with '&' as the escape character
static std::string quoted(const std::string& p) {
    std::string s = "\"";
    for (char ch : p) { if (ch == '"' || ch == '&') s += '&'; s += ch; }
    return s + "\"";
}

const char* distanceName(int which) {
    static const char* names[] = {"Manhattan", "Euclidean", "Cosine", "Canberra", "Jaccard"};
    return names[which];
}

// lib/include/kat/distance_metrics.hpp:39-127 (host floating point, printed with the stream's default precision)
double distanceMetric(int which, const std::vector<uint64_t>& s1, const std::vector<uint64_t>& s2) {
    const size_t n = s1.size();
    if (which == 0 || which == 1) {                               // Minkowski p = 1 / 2
        const int p = which == 0 ? 1 : 2;
        uint64_t sum = 0;
        for (size_t i = 0; i < n; i++) {
            uint64_t diff = s1[i] < s2[i] ? s2[i] - s1[i] : s1[i] - s2[i];
            sum += std::pow(diff, p);                             // uint64 += double, as written in the reference (:55)
        }
        return p == 1 ? (double)sum : std::pow((double)sum, 1.0 / (double)p);
    }
    if (which == 2) {                                             // Cosine
        double dot = 0.0, da = 0.0, db = 0.0;
        for (size_t i = 0; i < n; i++) { dot += s1[i] * s2[i]; da += std::pow(s1[i], 2); db += std::pow(s2[i], 2); }
        return 1.0 - (dot / (std::sqrt(da) * std::sqrt(db)));
    }
    if (which == 3) {                                             // Canberra
        double sum = 0.0;
        for (size_t i = 0; i < n; i++) {
            double diff = (double)s1[i] - (double)s2[i];
            double sum_i = s1[i] + s2[i];
            if (sum_i > 0) sum += std::abs(diff) / sum_i;
        }
        return sum;
    }
    double a = 0.0, b = 0.0;                                      // Jaccard
    for (size_t i = 0; i < n; i++) a += std::min(s1[i], s2[i]);
    for (size_t i = 0; i < n; i++) b += std::max(s1[i], s2[i]);
    return 1.0 - (a / b);
}

void CompCounters::printCounts(std::ostream& out) {              // lib/src/comp_counters.cc:144-206
    using std::endl;
    out << "K-mer statistics for: " << endl;
    out << " - Hash 1: " << quoted(hash1_path) << endl;
    out << " - Hash 2: " << quoted(hash2_path) << endl;
    if (hash3_total > 0) out << " - Hash 3: " << quoted(hash3_path) << endl;
    out << endl;
    out << "Total K-mers in: " << endl;
    out << " - Hash 1: " << hash1_total << endl;
    out << " - Hash 2: " << hash2_total << endl;
    if (hash3_total > 0) out << " - Hash 3: " << hash3_total << endl;
    out << endl;
    out << "Distinct K-mers in:" << endl;
    out << " - Hash 1: " << hash1_distinct << endl;
    out << " - Hash 2: " << hash2_distinct << endl;
    if (hash3_total > 0) out << " - Hash 3: " << hash3_distinct << endl;
    out << endl;
    out << "Total K-mers only found in:" << endl;
    out << " - Hash 1: " << hash1_only_total << endl;
    out << " - Hash 2: " << hash2_only_total << endl;
    out << endl;
    out << "Distinct K-mers only found in:" << endl;
    out << " - Hash 1: " << hash1_only_distinct << endl;
    out << " - Hash 2: " << hash2_only_distinct << endl << endl;
    out << "Shared K-mers:" << endl;
    out << " - Total shared found in hash 1: " << shared_hash1_total << endl;
    out << " - Total shared found in hash 2: " << shared_hash2_total << endl;
    out << " - Distinct shared K-mers: " << shared_distinct << endl << endl;
    out << "Distance between spectra 1 and 2 (all k-mers):" << endl;
    for (int i = 0; i < 5; i++) out << " - " << distanceName(i) << " distance: " << distanceMetric(i, spectrum1, spectrum2) << endl;
    out << endl;
    out << "Distance between spectra 1 and 2 (shared k-mers):" << endl;
    for (int i = 0; i < 5; i++) out << " - " << distanceName(i) << " distance: " << distanceMetric(i, shared_spectrum1, shared_spectrum2) << endl;
    out << endl;
}

// ---- option parsing (GNU style: --long value, --long=value, -s value, -svalue, switches; anything else positional) ----
bool ParsedArgs::has(const std::string& n) const {
    for (const auto& o : opts) if (o.first == n) return true;
    return false;
}
std::string ParsedArgs::get(const std::string& n, const std::string& def) const {
    std::string v = def;
    for (const auto& o : opts) if (o.first == n) v = o.second;
    return v;
}

ParsedArgs parseArgs(int argc, char* argv[], const std::vector<OptSpec>& spec) {
    ParsedArgs pa;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        const OptSpec* s = nullptr;
        std::string val;
        bool have_val = false;
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            std::string name = a.substr(2);
            size_t eq = name.find('=');
            if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); have_val = true; }
            for (const auto& o : spec) if (name == o.lng) s = &o;
            if (!s) throw OptionError("unrecognised option '" + a + "'");
        } else if (a.size() >= 2 && a[0] == '-' && !isdigit((unsigned char)a[1])) {
            for (const auto& o : spec) if (o.sht && a[1] == o.sht) s = &o;
            if (!s) throw OptionError("unrecognised option '" + a + "'");
            if (a.size() > 2) {
                if (s->takes_value) { val = a.substr(2); have_val = true; }
                else {                                          // bundled switches: -NO
                    for (size_t j = 1; j < a.size(); j++) {
                        const OptSpec* t = nullptr;
                        for (const auto& o : spec) if (o.sht && a[j] == o.sht) t = &o;
                        if (!t || t->takes_value) throw OptionError("unrecognised option '" + a + "'");
                        pa.opts.emplace_back(t->lng, "");
                    }
                    continue;
                }
            }
        } else {
            pa.positional.push_back(a);
            continue;
        }
        if (s->takes_value && !have_val) {
            if (i + 1 >= argc) throw OptionError(std::string("the required argument for option '--") + s->lng + "' is missing");
            val = argv[++i];
        }
        pa.opts.emplace_back(s->lng, val);
    }
    return pa;
}

std::vector<uint16_t> parseTrimList(const std::string& s) {      // boost::split(",") + lexical_cast<uint16_t>
    std::vector<uint16_t> out;
    std::stringstream ss(s);
    std::string tok;
    while (std::getline(ss, tok, ',')) {
        char* end = nullptr;
        long v = strtol(tok.c_str(), &end, 10);
        if (tok.empty() || *end || v < 0 || v > 65535) throw std::runtime_error("bad lexical cast: source type value could not be interpreted as target");
        out.push_back((uint16_t)v);
    }
    return out;
}

}  // namespace kat
