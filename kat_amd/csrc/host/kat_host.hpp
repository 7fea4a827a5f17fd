// kat_host.hpp -- C++ host side above the C ABI: the mirror of KAT's drivers for the hist / gcp / comp path.
//
// Same class and method names, argument meaning, defaults and error behaviour as the reference
// (src/histogram.{hpp,cc}, src/gcp.{hpp,cc}, src/comp.{hpp,cc}, lib/src/input_handler.cc, lib/src/comp_counters.cc,
// lib/include/kat/sparse_matrix.hpp), so a KAT maintainer reads it as KAT -- but InputHandler::count() and the
// bin/analyse/compare bodies are calls into libkatgpu.so (include/katgpu.h) instead of Jellyfish + std::thread teams.
// Links against the C ABI only: no HIP, no torch, nothing from oracle/.
#pragma once
#include <katgpu.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

namespace kat {

// KATGPU_TIMING=1: one "katgpu_timing {json}" line on stderr per phase of a run (process start -> device ready, each input's count,
// the reduction, the output files) next to the library's own per-file lines: what bench.py's end_to_end.breakdown is made of.
// Costs nothing otherwise; the KAT-format "Time taken" lines on stdout stay as they are (0.1 s resolution).
inline double timing_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline void timing_line(const char* phase, double ms, const char* what = "") {
    static const bool on = getenv("KATGPU_TIMING") != nullptr;
    if (!on) return;
    std::string w;                                        // (`what` may carry a path: quotes, backslashes and control characters escaped)
    for (const char* s = what; s && *s; ++s) {
        const unsigned char ch = (unsigned char)*s;
        if (ch == '"' || ch == '\\') { w += '\\'; w += (char)ch; }
        else if (ch < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", ch); w += b; }
        else w += (char)ch;
    }
    fprintf(stderr, "katgpu_timing {\"phase\": \"%s\", \"what\": \"%s\", \"ms\": %.1f}\n", phase, w.c_str(), ms);
}

const uint16_t DEFAULT_MER_LEN = 27;            // lib/include/kat/jellyfish_helper.hpp:76
const uint64_t DEFAULT_HASH_SIZE = 100000000;   // lib/include/kat/jellyfish_helper.hpp:75

// KAT's own exception family derives from boost::exception and makes `kat` exit with 4 (src/kat.cc:290-292);
// anything else derived from std::exception exits with 5 (src/kat.cc:293-295).
struct KatException : std::runtime_error {
    explicit KatException(const std::string& m) : std::runtime_error(m) {}
};
struct InputFileException : KatException { using KatException::KatException; };
struct JellyfishException : KatException { using KatException::KatException; };
struct HistogramException : KatException { using KatException::KatException; };
struct CompException : KatException { using KatException::KatException; };
struct FileSystemException : KatException { using KatException::KatException; };
struct SectException : KatException { using KatException::KatException; };

// One process-wide engine context (katgpu_init / katgpu_shutdown), and -- under `katgpu <mode> --gpus N` -- this process's place among
// the N that share the run: one per GPU, forked by kat_main.cc before any of them touches the device.  The reference has nothing to
// put beside this (one process, std::thread workers whose results it merges at the end: ThreadedSparseMatrix::mergeThreadedMatricies
// lib/include/kat/sparse_matrix.hpp:324-335, ThreadedCompCounters::merge lib/src/comp_counters.cc:230-254, Histogram::merge
// src/histogram.cc:146-160); here every rank counts its share of the reads, the tables are made one by owner over RCCL
// (katgpu_exchange_merge), the reducers run on the owned shards and their results are summed (katgpu_allreduce_u64).  Rank 0 speaks and
// writes the files; the other ranks stay silent.
class Engine {
public:
    static katgpu_ctx* ctx();
    static void check(int status);      // throws the exception the reference would have thrown
    static void shutdown();
    // multi-GPU
    static void setDist(int rank, int world, const std::string& id_file) { rank_ = rank; world_ = world; id_file_ = id_file; dist_ = true; }
    static bool dist() { return dist_; }            // --gpus was given (world 1 included: the protocol runs on the rank's own records)
    static int rank() { return rank_; }
    static int world() { return world_; }
    static bool speaker() { return rank_ == 0; }    // writes the output files
    static katgpu_comm* comm();                     // made on first use (rank 0 publishes the id through id_file)
    static void exchange(katgpu_table* t);          // no-op without --gpus
    static void exchangeBegin(katgpu_table* t);     // the same in two steps: the table's records travel while the caller counts its next input ...
    static bool exchangesUnderWay();
    static void finishPending();                    // ... and are applied here (katgpu_exchange_begin / _finish); every other collective finishes it first
    static void allreduce(uint64_t* buf, size_t n); // idem
    static void barrier();                          // idem
private:
    static int rank_, world_;
    static bool dist_;
    static std::string id_file_;
};

namespace mme {   // lib/include/kat/matrix_metadata_extractor.hpp:28-39
extern const char* const KEY_NB_COLUMNS; extern const char* const KEY_NB_ROWS; extern const char* const KEY_X_LABEL;
extern const char* const KEY_Y_LABEL; extern const char* const KEY_Z_LABEL; extern const char* const KEY_INPUT_1;
extern const char* const KEY_INPUT_2; extern const char* const KEY_KMER; extern const char* const KEY_TITLE;
extern const char* const KEY_MAX_VAL; extern const char* const KEY_TRANSPOSE; extern const char* const MX_META_END;
}

// Dense replacement for SparseMatrix<uint64_t> / ThreadedSparseMatrix (lib/include/kat/sparse_matrix.hpp): the device
// reduces into one dense uint64 array, so there are no per-thread maps to merge.
class Matrix64 {
public:
    Matrix64() = default;
    Matrix64(uint32_t rows, uint32_t cols) : m(rows), n(cols), v((size_t)rows * cols, 0) {}
    uint32_t width() const { return m; }      // reference naming: width() == number of rows (sparse_matrix.hpp:154-160)
    uint32_t height() const { return n; }
    uint64_t get(uint32_t i, uint32_t j) const { return v[(size_t)i * n + j]; }
    uint64_t* data() { return v.data(); }
    uint64_t getMaxVal() const;                 // sparse_matrix.hpp:162-173
    void printMatrix(std::ostream& out) const;  // sparse_matrix.hpp:255-279 (non-transposed form)
private:
    uint32_t m = 0, n = 0;
    std::vector<uint64_t> v;
};

// lib/include/kat/input_handler.hpp:33-79
class InputHandler {
public:
    enum InputMode { LOAD, COUNT };
    uint16_t index = 1;
    std::vector<std::string> input;
    std::vector<uint16_t> trim5p;
    InputMode mode = COUNT;
    bool canonical = false;
    uint64_t hashSize = DEFAULT_HASH_SIZE;
    uint16_t merLen = DEFAULT_MER_LEN;
    bool dumpHash = false;
    bool disableHashGrow = false;
    katgpu_table* hash = nullptr;               // was LargeHashArrayPtr

    ~InputHandler();
    void setSingleInput(const std::string& p) { input.assign(1, p); trim5p.assign(1, 0); }
    void setMultipleInputs(const std::vector<std::string>& inputs);
    std::string getSingleInput() const { return input[0]; }
    std::string pathString() const;
    std::string fileName() const;
    void set5pTrim(const std::vector<uint16_t>& trim_list);
    void validateInput();                        // throws if an input is missing; sets mode
    void count(uint16_t threads, const katgpu_table* like = nullptr, bool more_to_count = false);   // *** the drop-in boundary: katgpu_count ***
    void loadHeader() {}                         // the header travels with katgpu_jf_load (lib/src/input_handler.cc:139-143)
    void loadHash();                             // lib/src/input_handler.cc:204-219 -> katgpu_jf_load
    void validateMerLen(uint16_t merLen);        // lib/src/input_handler.cc:145-158
    void dump(const std::string& outputPath, uint16_t threads);   // lib/src/input_handler.cc:221-243 -> katgpu_jf_dump
    static std::shared_ptr<std::vector<std::string>> globFiles(const std::string& input);
    static std::shared_ptr<std::vector<std::string>> globFiles(const std::vector<std::string>& input);
    static bool isPipe(const std::string& p);
    static bool isSequenceFile(const std::string& p);
};

void ensureDirectoryExists(const std::string& dir);     // KatFS::ensureDirectoryExists (lib/include/kat/kat_fs.hpp:226-238)
std::string parentOfAbsolute(const std::string& prefix); // bfs::absolute(prefix).parent_path()

// src/histogram.hpp
class Histogram {
public:
    Histogram(const std::vector<std::string>& inputs, uint64_t low, uint64_t high, uint64_t inc);
    void setOutputPrefix(const std::string& p) { outputPrefix = p; }
    void setThreads(uint16_t t) { threads = t; }
    void setTrim(const std::vector<uint16_t>& t) { input.set5pTrim(t); }
    void setCanonical(bool c) { input.canonical = c; }
    void setMerLen(uint16_t m) { input.merLen = m; }
    void setHashSize(uint64_t h) { input.hashSize = h; }
    void setDumpHash(bool d) { input.dumpHash = d; }
    void setVerbose(bool v) { verbose = v; }
    void execute();
    void print(std::ostream& out);
    void save();
    static int main(int argc, char* argv[]);
    const std::vector<uint64_t>& getData() const { return data; }
private:
    uint64_t calcBase() const { return low > 1 ? low - 1 : 1; }   // src/histogram.hpp:172-174
    uint64_t calcCeil() const { return high + 1; }
    void bin();
    InputHandler input;
    std::string outputPrefix;
    uint64_t low, high, inc;
    uint16_t threads = 1;
    bool verbose = false;
    uint64_t base, ceil, nb_buckets;
    std::vector<uint64_t> data;
};

// src/gcp.hpp
class Gcp {
public:
    explicit Gcp(const std::vector<std::string>& inputs);
    void setOutputPrefix(const std::string& p) { outputPrefix = p; }
    void setThreads(uint16_t t) { threads = t; }
    void setTrim(const std::vector<uint16_t>& t) { input.set5pTrim(t); }
    void setCanonical(bool c) { input.canonical = c; }
    void setMerLen(uint16_t m) { input.merLen = m; }
    void setHashSize(uint64_t h) { input.hashSize = h; }
    void setDumpHash(bool d) { input.dumpHash = d; }
    void setCvgScale(double s) { cvgScale = s; }
    void setCvgBins(uint16_t b) { cvgBins = b; }
    void setVerbose(bool v) { verbose = v; }
    void execute();
    void printMainMatrix(std::ostream& out);
    void save();
    static int main(int argc, char* argv[]);
private:
    void analyse();
    InputHandler input;
    std::string outputPrefix;
    double cvgScale = 1.0;
    uint16_t cvgBins = 1000;
    uint16_t threads = 1;
    bool verbose = false;
    Matrix64 gcp_mx;
};

// lib/include/kat/comp_counters.hpp
class CompCounters {
public:
    uint64_t hash1_total = 0, hash2_total = 0, hash3_total = 0;
    uint64_t hash1_distinct = 0, hash2_distinct = 0, hash3_distinct = 0;
    uint64_t hash1_only_total = 0, hash2_only_total = 0, hash1_only_distinct = 0, hash2_only_distinct = 0;
    uint64_t shared_hash1_total = 0, shared_hash2_total = 0, shared_distinct = 0;
    std::vector<uint64_t> spectrum1, spectrum2, shared_spectrum1, shared_spectrum2;
    std::string hash1_path, hash2_path, hash3_path;

    CompCounters() : CompCounters("", "", "", 1001) {}
    CompCounters(const std::string& p1, const std::string& p2, const std::string& p3, size_t dm_size);
    void loadDevice(const uint64_t counters[13], const uint64_t* spectra);   // fills every field from katgpu_comp's outputs
    void printCounts(std::ostream& out);
    std::vector<uint64_t>& getSpectrum1() { return spectrum1; }
    std::vector<uint64_t>& getSpectrum2() { return spectrum2; }
};

// lib/include/kat/distance_metrics.hpp:39-127
double distanceMetric(int which, const std::vector<uint64_t>& s1, const std::vector<uint64_t>& s2);
const char* distanceName(int which);

// src/comp.hpp (two or three inputs)
class Comp {
public:
    Comp(const std::vector<std::string>& input1, const std::vector<std::string>& input2);
    void setThirdInput(const std::vector<std::string>& input3);       // src/comp.cc:100-106
    bool doThirdHash() const { return threeInputs; }
    size_t inputSize() const { return threeInputs ? 3 : 2; }
    void setOutputPrefix(const std::string& p) { outputPrefix = p; }
    void setD1Scale(double s) { d1Scale = s; }
    void setD2Scale(double s) { d2Scale = s; }
    void setD1Bins(uint16_t b) { d1Bins = b; }
    void setD2Bins(uint16_t b) { d2Bins = b; }
    void setThreads(uint16_t t) { threads = t; }
    void setMerLen(uint8_t m) { for (auto& in : input) in.merLen = m; }         // uint8_t, as in src/comp.hpp:167-171
    uint16_t getMerLen() const { return input[0].merLen; }
    void setTrim(size_t i, const std::vector<uint16_t>& t) { input[i].set5pTrim(t); }
    void setCanonical(size_t i, bool c) { input[i].canonical = c; }
    void setHashSize(size_t i, uint64_t h) { input[i].hashSize = h; }
    void setDumpHashes(bool d) { for (auto& in : input) in.dumpHash = d; }
    void setDisableHashGrow(bool d) { for (auto& in : input) in.disableHashGrow = d; }
    void setDensityPlot(bool d) { densityPlot = d; }
    void setOutputHists(bool h) { outputHists = h; }
    void setVerbose(bool v) { verbose = v; }
    void execute();
    void save();
    void printMainMatrix(std::ostream& out);
    void printEndsMatrix(std::ostream& out);
    void printMiddleMatrix(std::ostream& out);
    void printMixedMatrix(std::ostream& out);
    void printCounters(std::ostream& out) { comp_counters.printCounts(out); }
    void printHist(std::ostream& out, InputHandler& in, std::vector<uint64_t>& hist);
    static int main(int argc, char* argv[]);
private:
    void compare();
    InputHandler input[3];
    bool threeInputs = false;
    std::string outputPrefix;
    double d1Scale = 1.0, d2Scale = 1.0;
    uint16_t d1Bins = 1001, d2Bins = 1001;
    uint16_t threads = 1;
    bool densityPlot = false, outputHists = false, verbose = false;
    Matrix64 main_matrix, ends_matrix, middle_matrix, mixed_matrix;
    CompCounters comp_counters;
};

// FASTA / FASTQ (optionally gzip) records as seqan::SeqFileIn + readRecords(names, seqs, ...) deliver them to Sect
// (deps/seqan-library-2.0.0/include/seqan/seq_io/fasta_fastq.h:306-380): the name is the whole header line, only
// newlines are dropped from the sequence, and a sequence ends at the next '>' ('+' for FASTQ).  The format follows from the file
// name alone (.fa/.fasta, .fq/.fastq, .txt = one nameless record per line; optionally .gz); other names throw, as SeqAn does.
class SeqRecordReader {
public:
    explicit SeqRecordReader(const std::string& path);
    ~SeqRecordReader();
    bool atEnd();
    void readRecord(std::string& name, std::string& seq);
private:
    struct Impl;
    std::unique_ptr<Impl> impl;
    enum { FASTA, FASTQ, RAW } format = FASTA;
    int peek();
    void line(std::string* into);
};

// src/sect.hpp
class Sect {
public:
    Sect(const std::vector<std::string>& counts_files, const std::string& seq_file);
    void setOutputPrefix(const std::string& p) { outputPrefix = p; }
    void setGcBins(uint16_t b) { gcBins = b; }
    void setCvgBins(uint16_t b) { cvgBins = b; }
    void setCvgLogscale(bool l) { cvgLogscale = l; }
    void setThreads(uint16_t t) { threads = t; }
    void setTrim(const std::vector<uint16_t>& t) { input.set5pTrim(t); }
    void setCanonical(bool c) { input.canonical = c; }
    void setMerLen(uint16_t m) { input.merLen = m; }
    uint16_t getMerLen() const { return input.merLen; }
    void setHashSize(uint64_t h) { input.hashSize = h; }
    void setDumpHash(bool d) { input.dumpHash = d; }
    void setNoCountStats(bool n) { noCountStats = n; }
    void setOutputGCStats(bool g) { outputGCStats = g; }
    void setExtractNR(bool e) { extractNR = e; }
    void setExtractR(bool e) { extractR = e; }
    void setMinRepeat(uint32_t m) { minRepeat = m; }
    void setMaxRepeat(uint32_t m) { maxRepeat = m; }
    void setVerbose(bool v) { verbose = v; }
    void execute();
    void save();
    void printContaminationMatrix(std::ostream& out, const std::string& seq_file);
    static int main(int argc, char* argv[]);
private:
    struct Record {                         // the per-record slots of createBatchVars (src/sect.cc:310-322) + the text it prints
        const std::string* name = nullptr; const std::string* seq = nullptr;
        uint32_t median = 0, length = 0, invalid = 0, nonZero = 0;
        double mean = 0.0, gc = 0.0, percentInvalid = 0.0, percentNonZero = 0.0, percentNonZeroCorrected = 0.0;
        uint16_t mx_x = 0, mx_y = 0;
        std::string cvg_txt, gc_txt, nr_txt, r_txt;
    };
    void processSeqFile();
    void processSeq(Record& r, const uint64_t* counts);
    void regions(std::string& out, const Record& r, const uint64_t* counts, size_t nb, uint32_t min_count, uint32_t max_count);
    void merge();
    InputHandler input;
    std::string seqFile, outputPrefix;
    uint16_t gcBins = 1001, cvgBins = 1001;
    bool cvgLogscale = false;
    uint16_t threads = 1;
    bool noCountStats = false, outputGCStats = false, extractNR = false, extractR = false;
    uint32_t minRepeat = 2, maxRepeat = 0;
    bool verbose = false;
    Matrix64 contamination_mx;
};

// src/cold.hpp
class Cold {
public:
    Cold(const std::vector<std::string>& reads_files, const std::string& asm_file);
    void setOutputPrefix(const std::string& p) { outputPrefix = p; }
    void setReadsTrim(const std::vector<uint16_t>& t) { reads.set5pTrim(t); }
    void setCvgBins(uint16_t b) { cvgBins = b; }
    void setGcBins(uint16_t b) { gcBins = b; }
    void setThreads(uint16_t t) { threads = t; }
    void setHashSize(uint64_t h) { reads.hashSize = h; assembly.hashSize = h / 2; }         // src/cold.hpp:151-154
    void setMerLen(uint8_t m) { reads.merLen = m; assembly.merLen = m; }                    // uint8_t, src/cold.hpp:160-163
    uint8_t getMerLen() const { return (uint8_t)reads.merLen; }
    bool dumpHashes() const { return reads.dumpHash; }
    void setDumpHashes(bool d) { reads.dumpHash = d; assembly.dumpHash = d; }
    void setDisableHashGrow(bool d) { reads.disableHashGrow = d; }                          // src/cold.hpp:178-180
    void setVerbose(bool v) { verbose = v; }
    void execute();
    static int main(int argc, char* argv[]);
private:
    struct Row {
        uint32_t median = 0, asmCn = 0, length = 0, invalid = 0, nonZero = 0;
        double mean = 0.0, gc = 0.0, percentInvalid = 0.0, percentNonZero = 0.0, percentNonZeroCorrected = 0.0;
    };
    void processSeqFile();
    void processSeq(Row& r, const std::string& seq, const uint64_t* readsCounts, const uint64_t* asmCounts);
    InputHandler reads, assembly;
    std::string outputPrefix;
    uint16_t gcBins = 1001, cvgBins = 1001, threads = 1;
    bool verbose = false;
};

// ---- command-line helper shared by the three tools (stands in for boost::program_options) ----
struct OptSpec { const char* lng; char sht; bool takes_value; };
struct ParsedArgs {
    std::vector<std::pair<std::string, std::string>> opts;   // long name -> value ("" for switches)
    std::vector<std::string> positional;
    bool has(const std::string& n) const;
    std::string get(const std::string& n, const std::string& def) const;
};
struct OptionError : std::runtime_error { using std::runtime_error::runtime_error; };   // == po::error -> exit 1
ParsedArgs parseArgs(int argc, char* argv[], const std::vector<OptSpec>& spec);
std::vector<uint16_t> parseTrimList(const std::string& s);

}  // namespace kat
