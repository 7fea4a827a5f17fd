// tools.cc -- Histogram / Gcp / Comp: the three KAT drivers on the path (src/histogram.cc, src/gcp.cc, src/comp.cc).
// execute() keeps the reference's order of operations; bin() / analyse() / compare() are single calls into libkatgpu.
#include "kat_host.hpp"

#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>

using std::cout;
using std::endl;
using std::string;
using std::vector;

namespace kat {

namespace {
struct PhaseTimer {     // boost::timer::auto_cpu_timer(1, "  Time taken: %ws\n\n")
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    const char* fmt;
    const char* phase;  // KATGPU_TIMING: the phase's name in the "katgpu_timing" line
    explicit PhaseTimer(const char* f = "  Time taken: %.1fs\n\n", const char* ph = "reduce") : fmt(f), phase(ph) {}
    ~PhaseTimer() {
        char buf[128];
        timing_line(phase, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        snprintf(buf, sizeof buf, fmt, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        cout << buf;
        cout.flush();
    }
};
}  // namespace

// =========================================================== Histogram ===========================================

Histogram::Histogram(const vector<string>& inputs, uint64_t _low, uint64_t _high, uint64_t _inc) {   // src/histogram.cc:57-71
    input.setMultipleInputs(inputs);
    input.index = 1;
    outputPrefix = "kat-hist";
    low = _low; high = _high; inc = _inc;
    base = calcBase();
    ceil = calcCeil();
    nb_buckets = ceil + 1 - base;
}

void Histogram::execute() {                                                                       // src/histogram.cc:73-113
    if (high < low)
        throw HistogramException("High count value must be >= to low count value.  High: " + std::to_string(high) + "; Low: " + std::to_string(low));
    input.validateInput();
    ensureDirectoryExists(parentOfAbsolute(outputPrefix));
    if (input.mode == InputHandler::COUNT) input.count(threads);
    else { input.loadHeader(); input.loadHash(); }
    data.assign(nb_buckets, 0);
    bin();
    if (input.dumpHash) input.dump(outputPrefix + "-hash.jf" + std::to_string(input.merLen), threads);      // :105-108
    // merge(): nothing to do -- the device reduces into one array (the reference sums T per-thread histograms, :146-160)
}

void Histogram::bin() {                                                                           // src/histogram.cc:162-199
    PhaseTimer timer;
    cout << "Bining kmers ...";
    cout.flush();
    Engine::check(katgpu_hist(input.hash, base, ceil, inc, data.data(), nb_buckets));
    Engine::allreduce(data.data(), data.size());                 // Histogram::merge (src/histogram.cc:146-160), across GPUs
    cout << " done.";
    cout.flush();
}

void Histogram::print(std::ostream& out) {                                                        // src/histogram.cc:131-144
    out << mme::KEY_TITLE << input.merLen << "-mer spectra for: " << input.fileName() << endl;
    out << mme::KEY_X_LABEL << input.merLen << "-mer frequency" << endl;
    out << mme::KEY_Y_LABEL << "# distinct " << input.merLen << "-mers" << endl;
    out << mme::KEY_KMER << input.merLen << endl;
    out << mme::KEY_INPUT_1 << input.pathString() << endl;
    out << mme::MX_META_END << endl;
    uint64_t col = base;
    for (uint64_t i = 0; i < nb_buckets; i++, col += inc) out << col << " " << data[i] << "\n";
}

void Histogram::save() {                                                                          // src/histogram.cc:115-129
    if (!Engine::speaker()) return;                              // --gpus: rank 0 holds the whole result and writes it
    PhaseTimer timer;
    cout << "Saving results to disk ...";
    cout.flush();
    std::ofstream os(outputPrefix.c_str());
    print(os);
    os.close();
    cout << " done.";
    cout.flush();
}

int Histogram::main(int argc, char* argv[]) {                                                     // src/histogram.cc:257-370
    static const vector<OptSpec> spec = {
        {"output_prefix", 'o', true}, {"threads", 't', true}, {"low", 'l', true}, {"high", 'h', true}, {"inc", 'i', true},
        {"5ptrim", 0, true}, {"non_canonical", 'N', false}, {"mer_len", 'm', true}, {"hash_size", 'H', true},
        {"dump_hash", 'd', false}, {"output_type", 'p', true}, {"verbose", 'v', false}, {"help", 0, false}};
    ParsedArgs pa = parseArgs(argc, argv, spec);
    if (pa.has("help") || argc <= 1) {
        cout << "Usage: kat hist [options] (<input>)+\n\nCreate an histogram of k-mer occurrences from the input.\n" << endl;
        return 1;
    }
    vector<uint16_t> trim = parseTrimList(pa.get("5ptrim", "0"));
    PhaseTimer total("KAT HIST completed.\nTotal runtime: %.1fs\n\n", "total");
    cout << "Running KAT in HIST mode" << endl << "------------------------" << endl << endl;
    Histogram histo(pa.positional, std::stoull(pa.get("low", "1")), std::stoull(pa.get("high", "10000")), std::stoull(pa.get("inc", "1")));
    histo.setOutputPrefix(pa.get("output_prefix", "kat.hist"));
    histo.setThreads((uint16_t)std::stoul(pa.get("threads", "1")));
    histo.setTrim(trim);
    histo.setCanonical(!pa.has("non_canonical"));
    histo.setMerLen((uint16_t)std::stoul(pa.get("mer_len", std::to_string(DEFAULT_MER_LEN))));
    histo.setHashSize(std::stoull(pa.get("hash_size", std::to_string(DEFAULT_HASH_SIZE))));
    histo.setDumpHash(pa.has("dump_hash"));
    histo.setVerbose(pa.has("verbose"));
    histo.execute();
    { const double t0 = timing_now_ms(); histo.save(); timing_line("write_outputs", timing_now_ms() - t0); }
    return 0;
}

// =========================================================== Gcp =================================================

Gcp::Gcp(const vector<string>& inputs) {                                                          // src/gcp.cc:63-71
    input.setMultipleInputs(inputs);
    input.index = 1;
    outputPrefix = "kat-gcp";
}

void Gcp::execute() {                                                                             // src/gcp.cc:74-110
    input.validateInput();
    ensureDirectoryExists(parentOfAbsolute(outputPrefix));
    if (input.mode == InputHandler::COUNT) input.count(threads);
    else { input.loadHeader(); input.loadHash(); }
    // header->key_len() / 2 rows == k rows: GC count == k has no row (src/gcp.cc:93)
    gcp_mx = Matrix64(katgpu_table_k(input.hash), (uint32_t)cvgBins + 1);
    analyse();
    if (input.dumpHash) input.dump(outputPrefix + "-hash.jf" + std::to_string(input.merLen), threads);      // :102-105
}

void Gcp::analyse() {                                                                             // src/gcp.cc:158-197
    PhaseTimer timer;
    cout << "Analysing kmers in hash ...";
    cout.flush();
    Engine::check(katgpu_gcp(input.hash, cvgScale, cvgBins, gcp_mx.data()));
    Engine::allreduce(gcp_mx.data(), (size_t)gcp_mx.width() * gcp_mx.height());     // Gcp::merge (src/gcp.cc:128-138), across GPUs
    cout << "done.";
    cout.flush();
}

void Gcp::printMainMatrix(std::ostream& out) {                                                    // src/gcp.cc:140-156
    out << mme::KEY_TITLE << "K-mer coverage vs GC count plot for: " << input.fileName() << endl;
    out << mme::KEY_X_LABEL << input.merLen << "-mer frequency" << endl;
    out << mme::KEY_Y_LABEL << "GC count" << endl;
    out << mme::KEY_Z_LABEL << "# distinct " << input.merLen << "-mers" << endl;
    out << mme::KEY_NB_COLUMNS << gcp_mx.height() << endl;
    out << mme::KEY_NB_ROWS << gcp_mx.width() << endl;
    out << mme::KEY_MAX_VAL << gcp_mx.getMaxVal() << endl;
    out << mme::KEY_TRANSPOSE << "0" << endl;
    out << mme::KEY_KMER << input.merLen << endl;
    out << mme::KEY_INPUT_1 << input.pathString() << endl;
    out << mme::MX_META_END << endl;
    gcp_mx.printMatrix(out);
}

void Gcp::save() {                                                                                // src/gcp.cc:112-126
    if (!Engine::speaker()) return;                              // --gpus: rank 0 holds the whole result and writes it
    PhaseTimer timer;
    cout << "Saving results to disk ...";
    cout.flush();
    std::ofstream os((outputPrefix + ".mx").c_str());
    printMainMatrix(os);
    os.close();
    cout << " done.";
    cout.flush();
}

int Gcp::main(int argc, char* argv[]) {                                                           // src/gcp.cc:256-362
    static const vector<OptSpec> spec = {
        {"output_prefix", 'o', true}, {"threads", 't', true}, {"cvg_scale", 'x', true}, {"cvg_bins", 'y', true},
        {"5ptrim", 0, true}, {"non_canonical", 'N', false}, {"mer_len", 'm', true}, {"hash_size", 'H', true},
        {"dump_hash", 'd', false}, {"output_type", 'p', true}, {"verbose", 'v', false}, {"help", 0, false}};
    ParsedArgs pa = parseArgs(argc, argv, spec);
    if (pa.has("help") || argc <= 1) {
        cout << "Usage: kat gcp [options] (<input>)+\n\nCompares GC content and K-mer coverage from the input.\n" << endl;
        return 1;
    }
    vector<uint16_t> trim = parseTrimList(pa.get("5ptrim", "0"));
    PhaseTimer total("KAT GCP completed.\nTotal runtime: %.1fs\n\n", "total");
    cout << "Running KAT in GCP mode" << endl << "-----------------------" << endl << endl;
    Gcp gcp(pa.positional);
    gcp.setOutputPrefix(pa.get("output_prefix", "kat-gcp"));
    gcp.setThreads((uint16_t)std::stoul(pa.get("threads", "1")));
    gcp.setCanonical(!pa.has("non_canonical"));
    gcp.setCvgScale(std::stod(pa.get("cvg_scale", "1.0")));
    gcp.setCvgBins((uint16_t)std::stoul(pa.get("cvg_bins", "1000")));
    gcp.setTrim(trim);
    gcp.setMerLen((uint16_t)std::stoul(pa.get("mer_len", std::to_string(DEFAULT_MER_LEN))));
    gcp.setHashSize(std::stoull(pa.get("hash_size", std::to_string(DEFAULT_HASH_SIZE))));
    gcp.setDumpHash(pa.has("dump_hash"));
    gcp.setVerbose(pa.has("verbose"));
    gcp.execute();
    { const double t0 = timing_now_ms(); gcp.save(); timing_line("write_outputs", timing_now_ms() - t0); }
    return 0;
}

// =========================================================== Comp ================================================

Comp::Comp(const vector<string>& input1, const vector<string>& input2) {                          // src/comp.cc:73-98
    input[0].setMultipleInputs(input1);
    input[1].setMultipleInputs(input2);
    input[0].index = 1;
    input[1].index = 2;
    outputPrefix = "kat-comp";
}

void Comp::setThirdInput(const vector<string>& input3) {                                          // src/comp.cc:100-106
    input[2].setMultipleInputs(input3);
    input[2].index = 3;
    threeInputs = true;
}

void Comp::execute() {                                                                            // src/comp.cc:108-183
    for (size_t i = 0; i < inputSize(); i++) input[i].validateInput();
    ensureDirectoryExists(parentOfAbsolute(outputPrefix));
    main_matrix = Matrix64(d1Bins, d2Bins);
    if (doThirdHash()) { ends_matrix = Matrix64(d1Bins, d2Bins); middle_matrix = Matrix64(d1Bins, d2Bins); mixed_matrix = Matrix64(d1Bins, d2Bins); }
    comp_counters = CompCounters(input[0].getSingleInput(), input[1].getSingleInput(), doThirdHash() ? input[2].getSingleInput() : "",
                                 std::min(d1Bins, d2Bins));
    for (size_t i = 1; i < inputSize(); i++)                     // (the later inputs' tables: their memory is allocated while the first input is read)
        if (input[i].mode == InputHandler::COUNT && !(Engine::dist() && Engine::world() > 1)) katgpu_reserve(Engine::ctx(), getMerLen(), input[i].hashSize);
    for (size_t i = 0; i < inputSize(); i++) {                  // sequentially, one input after the other (:139-143)
        InputHandler& in = input[i];
        bool more = false;                                      // (--gpus N: this input's merge travels while the next one is counted)
        for (size_t j = i + 1; j < inputSize(); j++) more = more || input[j].mode == InputHandler::COUNT;
        if (in.mode == InputHandler::COUNT) in.count(threads, i > 0 ? input[0].hash : nullptr, more);
    }
    Engine::finishPending();
    bool anyLoad = false, allLoad = true;                        // :146-167
    for (size_t i = 0; i < inputSize(); i++) {
        if (input[i].mode == InputHandler::LOAD) { input[i].loadHeader(); anyLoad = true; }
        else allLoad = false;
    }
    if (anyLoad)
        for (size_t i = 0; i < inputSize(); i++) if (input[i].mode == InputHandler::LOAD) input[i].loadHash();
    if (allLoad) setMerLen((uint8_t)katgpu_table_k(input[0].hash));
    for (size_t i = 0; i < inputSize(); i++) input[i].validateMerLen(getMerLen());
    compare();
    if (input[0].dumpHash)                                       // :174-179
        for (size_t i = 0; i < inputSize(); i++)
            input[i].dump(outputPrefix + "-hash" + std::to_string(input[i].index) + ".jf" + std::to_string(getMerLen()), threads);
    // merge(): the reference's dense T-way map merge (:248-265) has no counterpart, the device produced one matrix
}

void Comp::compare() {                                                                            // src/comp.cc:366-385
    PhaseTimer timer;
    cout << "Comparing hashes ...";
    cout.flush();
    const uint32_t ss = std::min(d1Bins, d2Bins);
    uint64_t counters[13];
    vector<uint64_t> spectra((size_t)4 * ss);
    if (doThirdHash())
        Engine::check(katgpu_comp3(input[0].hash, input[1].hash, input[2].hash, input[0].canonical, input[1].canonical, input[2].canonical,
                                   d1Scale, d2Scale, d1Bins, d2Bins, main_matrix.data(), ends_matrix.data(), middle_matrix.data(),
                                   mixed_matrix.data(), counters, spectra.data()));
    else
        Engine::check(katgpu_comp(input[0].hash, input[1].hash, input[0].canonical, input[1].canonical, d1Scale, d2Scale, d1Bins, d2Bins,
                                  main_matrix.data(), counters, spectra.data()));
    if (Engine::dist()) {                                        // Comp::merge (src/comp.cc:248-265) + ThreadedCompCounters::merge, across GPUs
        Engine::allreduce(main_matrix.data(), (size_t)d1Bins * d2Bins);
        if (doThirdHash()) { Engine::allreduce(ends_matrix.data(), (size_t)d1Bins * d2Bins); Engine::allreduce(middle_matrix.data(), (size_t)d1Bins * d2Bins); Engine::allreduce(mixed_matrix.data(), (size_t)d1Bins * d2Bins); }
        Engine::allreduce(counters, 13);
        Engine::allreduce(spectra.data(), spectra.size());
    }
    comp_counters.loadDevice(counters, spectra.data());
    cout << " done.";
    cout.flush();
}

void Comp::printMainMatrix(std::ostream& out) {                                                   // src/comp.cc:308-326
    out << mme::KEY_TITLE << "K-mer comparison plot" << endl
        << mme::KEY_X_LABEL << input[0].merLen << "-mer frequency for: " << input[0].fileName() << endl
        << mme::KEY_Y_LABEL << input[1].merLen << "-mer frequency for: " << input[1].fileName() << endl
        << mme::KEY_Z_LABEL << "# distinct " << input[0].merLen << "-mers" << endl
        << mme::KEY_NB_COLUMNS << main_matrix.height() << endl
        << mme::KEY_NB_ROWS << main_matrix.width() << endl
        << mme::KEY_MAX_VAL << main_matrix.getMaxVal() << endl
        << mme::KEY_TRANSPOSE << "1" << endl
        << mme::KEY_KMER << input[0].merLen << endl
        << mme::KEY_INPUT_1 << input[0].pathString() << endl
        << mme::KEY_INPUT_2 << input[1].pathString() << endl
        << mme::MX_META_END << endl;
    main_matrix.printMatrix(out);
}

void Comp::printEndsMatrix(std::ostream& out) {                                                   // src/comp.cc:330-337
    out << "# Each row represents K-mer frequency for: " << input[0].getSingleInput() << endl;
    out << "# Each column represents K-mer frequency for sequence ends: " << input[2].getSingleInput() << endl;
    ends_matrix.printMatrix(out);
}

void Comp::printMiddleMatrix(std::ostream& out) {                                                 // src/comp.cc:341-348
    out << "# Each row represents K-mer frequency for: " << input[0].getSingleInput() << endl;
    out << "# Each column represents K-mer frequency for sequence middles: " << input[1].getSingleInput() << endl;
    middle_matrix.printMatrix(out);
}

void Comp::printMixedMatrix(std::ostream& out) {                                                  // src/comp.cc:352-358
    out << "# Each row represents K-mer frequency for hash file 1: " << input[0].getSingleInput() << endl;
    out << "# Each column represents K-mer frequency for mixed: " << input[1].getSingleInput() << " and " << input[2].getSingleInput() << endl;
    mixed_matrix.printMatrix(out);
}

void Comp::printHist(std::ostream& out, InputHandler& in, vector<uint64_t>& hist) {               // src/comp.cc:235-246
    out << mme::KEY_TITLE << in.merLen << "-mer spectra for: " << in.pathString() << endl;
    out << mme::KEY_X_LABEL << in.merLen << "-mer frequency" << endl;
    out << mme::KEY_Y_LABEL << "# distinct " << in.merLen << "-mers" << endl;
    out << mme::MX_META_END << endl;
    for (uint64_t i = 0; i < hist.size(); i++) out << i << " " << hist[i] << "\n";
}

void Comp::save() {                                                                               // src/comp.cc:185-233
    if (!Engine::speaker()) return;                              // --gpus: rank 0 holds the whole result and writes it
    PhaseTimer timer;
    cout << "Saving results to disk ...";
    cout.flush();
    std::ofstream mx((outputPrefix + "-main.mx").c_str());
    printMainMatrix(mx);
    mx.close();
    if (doThirdHash()) {                                                                          // src/comp.cc:197-213
        std::ofstream e((outputPrefix + "-ends.mx").c_str());
        printEndsMatrix(e);
        e.close();
        std::ofstream m((outputPrefix + "-middle.mx").c_str());
        printMiddleMatrix(m);
        m.close();
        std::ofstream x((outputPrefix + "-mixed.mx").c_str());
        printMixedMatrix(x);
        x.close();
    }
    std::ofstream st((outputPrefix + ".stats").c_str());
    printCounters(st);
    st.close();
    if (outputHists) {
        std::ofstream h1((outputPrefix + ".1.hist").c_str());
        printHist(h1, input[0], comp_counters.getSpectrum1());
        h1.close();
        std::ofstream h2((outputPrefix + ".2.hist").c_str());
        printHist(h2, input[1], comp_counters.getSpectrum2());
        h2.close();
    }
    cout << " done.";
    cout.flush();
}

int Comp::main(int argc, char* argv[]) {                                                          // src/comp.cc:633-845
    static const vector<OptSpec> spec = {
        {"output_prefix", 'o', true}, {"threads", 't', true}, {"d1_scale", 'x', true}, {"d2_scale", 'y', true},
        {"d1_bins", 'i', true}, {"d2_bins", 'j', true}, {"d1_5ptrim", 0, true}, {"d2_5ptrim", 0, true},
        {"non_canonical_1", 'N', false}, {"non_canonical_2", 'O', false}, {"non_canonical_3", 'P', false},
        {"mer_len", 'm', true}, {"hash_size_1", 'H', true}, {"hash_size_2", 'I', true}, {"hash_size_3", 'J', true},
        {"dump_hashes", 'd', false}, {"disable_hash_grow", 'g', false}, {"density_plot", 'n', false},
        {"output_type", 'p', true}, {"output_hists", 'h', false}, {"verbose", 'v', false}, {"help", 0, false}};
    ParsedArgs pa = parseArgs(argc, argv, spec);
    if (pa.has("help") || argc <= 1) {
        cout << "Usage: kat comp [options] <input_1> <input_2> [<input_3>]\n\nCompares jellyfish K-mer count hashes.\n" << endl;
        return 1;
    }
    PhaseTimer total("KAT COMP completed.\nTotal runtime: %.1fs\n\n", "total");
    cout << "Running KAT in COMP mode" << endl << "------------------------" << endl << endl;
    const bool verbose = pa.has("verbose");
    if (pa.positional.empty() || pa.positional[0].empty()) throw CompException("Nothing specified for input group 1");
    if (verbose) std::cerr << "Input 1: " << pa.positional[0] << endl << endl;
    auto vec1 = InputHandler::globFiles(pa.positional[0]);
    if (pa.positional.size() < 2 || pa.positional[1].empty()) throw CompException("Nothing specified for input group 2");
    if (verbose) std::cerr << "Input 2: " << pa.positional[1] << endl << endl;
    auto vec2 = InputHandler::globFiles(pa.positional[1]);
    Comp comp(*vec1, *vec2);
    if (pa.positional.size() > 2 && !pa.positional[2].empty()) {
        if (verbose) std::cerr << "Input 3: " << pa.positional[2] << endl << endl;
        comp.setThirdInput(*InputHandler::globFiles(pa.positional[2]));
    }
    comp.setOutputPrefix(pa.get("output_prefix", "kat-comp"));
    comp.setD1Scale(std::stod(pa.get("d1_scale", "1.0")));
    comp.setD2Scale(std::stod(pa.get("d2_scale", "1.0")));
    comp.setTrim(0, parseTrimList(pa.get("d1_5ptrim", "0")));
    comp.setTrim(1, parseTrimList(pa.get("d2_5ptrim", "0")));
    comp.setD1Bins((uint16_t)std::stoul(pa.get("d1_bins", "1001")));
    comp.setD2Bins((uint16_t)std::stoul(pa.get("d2_bins", "1001")));
    comp.setThreads((uint16_t)std::stoul(pa.get("threads", "1")));
    comp.setMerLen((uint8_t)std::stoul(pa.get("mer_len", std::to_string(DEFAULT_MER_LEN))));
    comp.setCanonical(0, !pa.has("non_canonical_1"));
    comp.setCanonical(1, !pa.has("non_canonical_2"));
    comp.setCanonical(2, !pa.has("non_canonical_3"));
    comp.setHashSize(2, std::stoull(pa.get("hash_size_3", std::to_string(DEFAULT_HASH_SIZE))));
    comp.setHashSize(0, std::stoull(pa.get("hash_size_1", std::to_string(DEFAULT_HASH_SIZE))));
    comp.setHashSize(1, std::stoull(pa.get("hash_size_2", std::to_string(DEFAULT_HASH_SIZE))));
    comp.setDumpHashes(pa.has("dump_hashes"));
    comp.setDisableHashGrow(pa.has("disable_hash_grow"));
    comp.setDensityPlot(pa.has("density_plot"));
    comp.setOutputHists(pa.has("output_hists"));
    comp.setVerbose(verbose);
    comp.execute();
    { const double t0 = timing_now_ms(); comp.save(); timing_line("write_outputs", timing_now_ms() - t0); }
    cout << endl << "Summary statistics" << endl << "------------------" << endl << endl;
    comp.printCounters(cout);
    return 0;
}

}  // namespace kat
