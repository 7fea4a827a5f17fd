// kat_ref_parts.cc -- driver around the parts of KAT 2.4.2's own library that compile without its generated config.h
// (test infrastructure).  Compiled by oracle/Makefile (`make ref`) against lib/include/kat/{comp_counters,distance_metrics,
// sparse_matrix,str_utils}.hpp and lib/src/comp_counters.cc where they lie, with the vendored boost headers (deps/boost,
// boost.system header-only), into oracle/_ref/kat_ref_parts.  The drivers of the tools (src/*.cc, lib/src/input_handler.cc,
// jellyfish_helper.cc) include <config.h> unconditionally and are not buildable here.
//
//   kat_ref_parts compstats           stdin: path1\npath2\npath3\nN\n13 counters\n4 x N spectra  ->  CompCounters::printCounts
//   kat_ref_parts compupdate          stdin: N, then lines "c1 c2 [c3]" applied the way Comp::compareSlice applies them
//                                     (src/comp.cc:401-433: hash-1 walk, then hash-2 walk for k-mers absent from hash 1)
//   kat_ref_parts distance            stdin: N, spectrum a, spectrum b  ->  the five metrics, default stream precision
//   kat_ref_parts matrix R C          stdin: "i j v" triples (inc)  ->  getMaxVal, then printMatrix
//   kat_ref_parts strutils            stdin: one string per line  ->  "validKmer gcCount"
//   kat_ref_parts mxread <file.mx>    the reference's own readers on a matrix file: matrix_metadata_extractor (every key) and
//                                     SparseMatrix(path): rows, columns, getMaxVal, sum of all cells
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <boost/exception/all.hpp>

// (the order lib/src/comp_counters.cc includes them in: distance_metrics.hpp relies on using-declarations made before it)
#include <kat/str_utils.hpp>
#include <kat/sparse_matrix.hpp>
#include <kat/comp_counters.hpp>
#include <kat/distance_metrics.hpp>
#include <kat/matrix_metadata_extractor.hpp>

using std::cin;
using std::cout;
using std::endl;

int main(int argc, char* argv[]) {
    if (argc < 2) return 2;
    const std::string mode = argv[1];
    if (mode == "compstats") {
        std::string p1, p2, p3;
        std::getline(cin, p1); std::getline(cin, p2); std::getline(cin, p3);
        size_t n; cin >> n;
        kat::CompCounters cc(p1, p2, p3, n);
        cin >> cc.hash1_total >> cc.hash2_total >> cc.hash3_total >> cc.hash1_distinct >> cc.hash2_distinct >> cc.hash3_distinct >> cc.hash1_only_total >>
            cc.hash2_only_total >> cc.hash1_only_distinct >> cc.hash2_only_distinct >> cc.shared_hash1_total >> cc.shared_hash2_total >> cc.shared_distinct;
        for (size_t i = 0; i < n; ++i) cin >> cc.spectrum1[i];
        for (size_t i = 0; i < n; ++i) cin >> cc.spectrum2[i];
        for (size_t i = 0; i < n; ++i) cin >> cc.shared_spectrum1[i];
        for (size_t i = 0; i < n; ++i) cin >> cc.shared_spectrum2[i];
        cc.printCounts(cout);
        return 0;
    }
    if (mode == "compupdate") {
        size_t n; cin >> n;
        kat::CompCounters cc("1", "2", "", n);
        std::string tag; uint64_t a, b;
        while (cin >> tag >> a >> b) {
            if (tag == "h1") { cc.updateHash1Counters(a, b); cc.updateSharedCounters(a, b); }      // every k-mer of hash 1 (src/comp.cc:420-425)
            else if (tag == "h2") cc.updateHash2Counters(a, b);                                   // every k-mer of hash 2 (src/comp.cc:455-465)
        }
        cout << cc.hash1_total << ' ' << cc.hash2_total << ' ' << cc.hash3_total << ' ' << cc.hash1_distinct << ' ' << cc.hash2_distinct << ' ' << cc.hash3_distinct << ' '
             << cc.hash1_only_total << ' ' << cc.hash2_only_total << ' ' << cc.hash1_only_distinct << ' ' << cc.hash2_only_distinct << ' ' << cc.shared_hash1_total << ' '
             << cc.shared_hash2_total << ' ' << cc.shared_distinct << endl;
        for (size_t i = 0; i < n; ++i) cout << cc.spectrum1[i] << (i + 1 < n ? ' ' : '\n');
        for (size_t i = 0; i < n; ++i) cout << cc.spectrum2[i] << (i + 1 < n ? ' ' : '\n');
        for (size_t i = 0; i < n; ++i) cout << cc.shared_spectrum1[i] << (i + 1 < n ? ' ' : '\n');
        for (size_t i = 0; i < n; ++i) cout << cc.shared_spectrum2[i] << (i + 1 < n ? ' ' : '\n');
        return 0;
    }
    if (mode == "distance") {
        size_t n; cin >> n;
        std::vector<uint64_t> a(n), b(n);
        for (size_t i = 0; i < n; ++i) cin >> a[i];
        for (size_t i = 0; i < n; ++i) cin >> b[i];
        std::vector<std::unique_ptr<kat::DistanceMetric>> dms;
        dms.push_back(std::unique_ptr<kat::DistanceMetric>(new kat::ManhattanDistance()));
        dms.push_back(std::unique_ptr<kat::DistanceMetric>(new kat::EuclideanDistance()));
        dms.push_back(std::unique_ptr<kat::DistanceMetric>(new kat::CosineDistance()));
        dms.push_back(std::unique_ptr<kat::DistanceMetric>(new kat::CanberraDistance()));
        dms.push_back(std::unique_ptr<kat::DistanceMetric>(new kat::JaccardDistance()));
        for (auto& dm : dms) cout << dm->calcDistance(a, b) << endl;
        return 0;
    }
    if (mode == "matrix" && argc >= 4) {
        kat::SM64 m(atoi(argv[2]), atoi(argv[3]));
        uint32_t i, j; uint64_t v;
        while (cin >> i >> j >> v) m.inc(i, j, v);
        cout << m.getMaxVal() << endl;
        m.printMatrix(cout);
        return 0;
    }
    if (mode == "mxread" && argc >= 3) {
        const path p(argv[2]);
        cout << mme::getNumeric(p, mme::KEY_NB_COLUMNS) << ' ' << mme::getNumeric(p, mme::KEY_NB_ROWS) << ' ' << mme::getNumeric(p, mme::KEY_MAX_VAL) << ' '
             << mme::getNumeric(p, mme::KEY_TRANSPOSE) << ' ' << mme::getNumeric(p, mme::KEY_KMER) << endl;
        cout << mme::getString(p, mme::KEY_TITLE) << endl << mme::getString(p, mme::KEY_X_LABEL) << endl << mme::getString(p, mme::KEY_Y_LABEL) << endl
             << mme::getString(p, mme::KEY_Z_LABEL) << endl << mme::getString(p, mme::KEY_INPUT_1) << endl << mme::getString(p, mme::KEY_INPUT_2) << endl;
        kat::SM64 m(p);
        unsigned long long sum = 0;
        for (uint32_t i = 0; i < m.height(); ++i) for (uint32_t j = 0; j < m.width(); ++j) sum += m.get(j, i);
        cout << m.width() << ' ' << m.height() << ' ' << m.getMaxVal() << ' ' << sum << endl;
        return 0;
    }
    if (mode == "strutils") {
        std::string line;
        while (std::getline(cin, line)) cout << kat::validKmer(line) << ' ' << kat::gcCount(line) << endl;
        return 0;
    }
    return 2;
}
