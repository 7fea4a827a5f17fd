// sect.cc -- `kat sect` (src/sect.cc): k-mer coverage along every record of a sequence file.
//
// The reference walks each record with substr + validKmer + mer_dna + JellyfishHelper::getCount (src/sect.cc:516-535), one
// record per std::thread.  Here a batch of records is joined into one base buffer and profiled by a single
// katgpu_table_profile_host call; what stays on the host is what the reference also does after the lookups: the
// per-record statistics and the text.  Records are read with the semantics of the vendored SeqAn 2.0.0 reader
// (deps/seqan-library-2.0.0/include/seqan/seq_io/fasta_fastq.h:306-380, CharString target).
#include "kat_host.hpp"

#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <thread>

using std::cout;
using std::endl;
using std::string;
using std::vector;

namespace kat {

namespace {
struct PhaseTimer {     // boost::timer::auto_cpu_timer(1, "  Time taken: %ws\n\n")
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    const char* fmt;
    explicit PhaseTimer(const char* f = "  Time taken: %.1fs\n\n") : fmt(f) {}
    ~PhaseTimer() {
        char buf[128];
        snprintf(buf, sizeof buf, fmt, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        cout << buf;
        cout.flush();
    }
};

bool endsWithNoCase(const string& s, const char* suffix) {
    const size_t n = strlen(suffix);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; i++) if (tolower((unsigned char)s[s.size() - n + i]) != suffix[i]) return false;
    return true;
}

inline bool isBase(char c) {                    // lib/include/kat/str_utils.hpp:183-201 (validKmer)
    switch (c) { case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': return true; default: return false; }
}
inline bool isGC(char c) { return c == 'G' || c == 'g' || c == 'C' || c == 'c'; }

inline void appendU64(string& s, uint64_t v) {
    char tmp[24]; int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) s.push_back(tmp[--n]);
}
}  // namespace

// ---- SeqRecordReader: FASTA / FASTQ records the way seqan::SeqFileIn + readRecords deliver them ----
struct SeqRecordReader::Impl {
    gzFile f = nullptr;
    vector<unsigned char> buf = vector<unsigned char>(1 << 20);
    size_t pos = 0, len = 0;
    bool fill() {
        if (pos < len) return true;
        int r = gzread(f, buf.data(), (unsigned)buf.size());
        if (r <= 0) { len = pos = 0; return false; }
        len = (size_t)r; pos = 0;
        return true;
    }
};

SeqRecordReader::SeqRecordReader(const string& path) : impl(new Impl) {
    impl->f = gzopen(path.c_str(), "rb");
    if (!impl->f) throw std::runtime_error("Could not open sequence file: " + path);
    string base = path;
    if (endsWithNoCase(base, ".gz")) base.resize(base.size() - 3);
    // SeqAn decides on the file name alone (fasta_fastq.h:104-143), case-insensitively; anything else is UnknownExtensionError
    if (endsWithNoCase(base, ".fa") || endsWithNoCase(base, ".fasta")) format = FASTA;
    else if (endsWithNoCase(base, ".fq") || endsWithNoCase(base, ".fastq")) format = FASTQ;
    else if (endsWithNoCase(base, ".txt")) format = RAW;
    else throw std::runtime_error("Unknown file extension of " + path + ": iostream error");
}

SeqRecordReader::~SeqRecordReader() { if (impl->f) gzclose(impl->f); }

int SeqRecordReader::peek() { return impl->fill() ? impl->buf[impl->pos] : -1; }
bool SeqRecordReader::atEnd() { return !impl->fill(); }

void SeqRecordReader::line(string* into) {      // readLine / skipLine (seqan/stream/tokenization.h:408-455)
    int c;
    while ((c = peek()) >= 0 && c != '\n' && c != '\r') { if (into) into->push_back((char)c); impl->pos++; }
    if (peek() == '\r') impl->pos++;
    if (peek() == '\n') impl->pos++;
}

void SeqRecordReader::readRecord(string& name, string& seq) {
    name.clear(); seq.clear();
    if (format == RAW) { line(&seq); return; }                                                  // every line a nameless record
    const bool fastq = format == FASTQ;
    const int begin = fastq ? '@' : '>', stop = fastq ? '+' : '>';
    int c;
    while ((c = peek()) >= 0 && c != begin) impl->pos++;
    if (c < 0) throw std::runtime_error("Unexpected end of input.");                            // seqan::UnexpectedEnd
    impl->pos++;
    line(&name);
    // the sequence runs to the next '>' ('+' in FASTQ) wherever it stands; for a char target only newlines are dropped
    while (impl->fill()) {
        unsigned char* p = impl->buf.data() + impl->pos;
        unsigned char* e = impl->buf.data() + impl->len;
        unsigned char* q = p;
        while (q < e && *q != stop && *q != '\n' && *q != '\r') q++;
        seq.append((const char*)p, (size_t)(q - p));
        impl->pos += (size_t)(q - p);
        if (q == e) continue;
        if (*q == stop) break;
        impl->pos++;
    }
    if (!fastq) return;
    if (peek() != '+') throw std::runtime_error("Unexpected end of input.");
    impl->pos++;
    line(nullptr);                                                                              // optional second id
    size_t left = seq.size();                                                                   // CountDownFunctor over non-newline characters
    while (left && (c = peek()) >= 0) { if (c != '\n' && c != '\r') left--; impl->pos++; }
    while ((c = peek()) >= 0 && c != '@') impl->pos++;
}

// ---- Sect ----
Sect::Sect(const vector<string>& counts_files, const string& seq_file) {                        // src/sect.cc:65-83
    input.setMultipleInputs(counts_files);
    input.index = 1;
    seqFile = seq_file;
    outputPrefix = "kat-sect";
}

void Sect::execute() {                                                                          // src/sect.cc:86-125
    struct stat st;
    if (lstat(seqFile.c_str(), &st) != 0)
        throw SectException("Could not find sequence file at: " + seqFile + "; please check the path and try again.");
    input.validateInput();
    ensureDirectoryExists(parentOfAbsolute(outputPrefix));
    if (input.mode == InputHandler::COUNT) input.count(threads);
    else { input.loadHeader(); input.loadHash(); }
    contamination_mx = Matrix64(gcBins, cvgBins);
    processSeqFile();
    if (input.dumpHash) input.dump(outputPrefix + "-hash.jf" + std::to_string(input.merLen), threads);
    merge();
}

void Sect::merge() {                                                                            // src/sect.cc:246-256
    PhaseTimer timer;
    cout << "Merging matrices ...";     // nothing to merge: one matrix, filled on the host from per-record scalars
    cout << " done.";
    cout.flush();
}

void Sect::save() {                                                                             // src/sect.cc:127-141 (never called by Sect::main)
    PhaseTimer timer;
    cout << "Saving results to disk ...";
    cout.flush();
    std::ofstream os((outputPrefix + "-contamination.mx").c_str());
    printContaminationMatrix(os, seqFile);
    os.close();
    cout << " done.";
    cout.flush();
}

void Sect::printContaminationMatrix(std::ostream& out, const string& seq_file) {               // src/sect.cc:449-463
    out << mme::KEY_TITLE << "Contamination Plot for " << seq_file << " and " << "\"\"" << endl;   // hashFile is never set: boost prints ""
    out << mme::KEY_X_LABEL << "GC%" << endl;
    out << mme::KEY_Y_LABEL << "Average K-mer Coverage" << endl;
    out << mme::KEY_Z_LABEL << "Base Count per bin" << endl;
    out << mme::KEY_NB_COLUMNS << gcBins << endl;
    out << mme::KEY_NB_ROWS << cvgBins << endl;
    out << mme::KEY_MAX_VAL << contamination_mx.getMaxVal() << endl;
    out << mme::KEY_TRANSPOSE << "0" << endl;
    out << mme::MX_META_END << endl;
    contamination_mx.printMatrix(out);
}

// Everything the reference's processSeq + print* produce for one record, from the device's per-position counts.
void Sect::processSeq(Record& r, const uint64_t* cnt) {                                         // src/sect.cc:486-603
    const uint16_t k = input.merLen;
    const string& seq = *r.seq;
    const uint64_t seqLength = seq.size();
    const int64_t nbCounts = (int64_t)seqLength - k + 1;
    uint64_t nbNonZero = 0, nbInvalid = 0;
    r.median = 0; r.mean = 0.0;
    const size_t nb = nbCounts > 0 ? (size_t)nbCounts : 0;
    vector<int16_t> gc;
    if (nb) {
        // validity and GC of every window by a rolling scan (the reference re-reads k characters per window)
        gc.resize(nb);
        uint32_t bad = 0, g = 0;
        for (size_t i = 0; i < seqLength; i++) {
            bad += !isBase(seq[i]); g += isGC(seq[i]);
            if (i >= k) { bad -= !isBase(seq[i - k]); g -= isGC(seq[i - k]); }
            if (i + 1 >= k) gc[i + 1 - k] = bad ? (int16_t)-1 : (int16_t)g;
        }
        uint64_t sum = 0;
        for (size_t i = 0; i < nb; i++) {
            if (gc[i] < 0) nbInvalid++;
            else { sum += cnt[i]; if (cnt[i]) nbNonZero++; }
        }
        vector<uint64_t> sorted(cnt, cnt + nb);
        std::nth_element(sorted.begin(), sorted.begin() + nb / 2, sorted.end());                // == sort()[size/2] (:540-542)
        r.median = (uint32_t)(double)sorted[nb / 2];
        r.mean = (double)sum / (double)nbCounts;
    }
    r.length = (uint32_t)seqLength;
    r.nonZero = (uint32_t)nbNonZero;
    r.percentNonZero = nbNonZero == 0 || nbCounts <= 0 ? 0.0 : ((double)nbNonZero / (double)nbCounts) * 100.0;
    r.invalid = (uint32_t)nbInvalid;
    r.percentInvalid = nbInvalid == 0 || nbCounts <= 0 ? 0.0 : ((double)nbInvalid / (double)nbCounts) * 100.0;
    const uint64_t notInvalid = (uint64_t)nbCounts - nbInvalid;
    r.percentNonZeroCorrected = nbNonZero == 0 || notInvalid <= 0 ? 0.0 : ((double)nbNonZero / (double)notInvalid) * 100.0;

    uint64_t gs = 0, cs = 0, ns = 0;
    for (char c : seq) {
        if (c == 'G' || c == 'g') gs++;
        else if (c == 'C' || c == 'c') cs++;
        else if (c == 'N' || c == 'n') ns++;
    }
    volatile double num = (double)(gs + cs), den = (double)(seqLength - ns);                    // run-time division: 0/0 is the x86 default NaN ("-nan")
    r.gc = num / den;

    // src/sect.cc:581-600.  average_cvg is never assigned there, so the coverage bin is 0 with or without
    // --cvg_logscale (uint16_t(-inf) is 0 on x86, like uint16_t(NaN) for the GC bin of an all-N record).
    const double xd = r.gc * gcBins;
    r.mx_x = std::isnan(xd) ? 0 : (uint16_t)xd;
    r.mx_y = 0;

    if (!noCountStats) {                                                                        // printCounts, :328-346
        string& o = r.cvg_txt;
        o.reserve(nb * 2 + r.name->size() + 8);
        o += '>'; o += *r.name; o += '\n';
        if (nb) {
            appendU64(o, cnt[0]);
            for (size_t j = 1; j < nb; j++) { o += ' '; appendU64(o, cnt[j]); }
            o += '\n';
        } else o += "0\n";
    }
    if (outputGCStats) {                                                                        // printGCCounts, :352-371
        vector<string> pct(k + 1);
        char tmp[32];
        for (uint16_t g = 0; g <= k; g++) { snprintf(tmp, sizeof tmp, "%.1f", ((double)g / (double)k) * 100.0); pct[g] = tmp; }
        string& o = r.gc_txt;
        o += '>'; o += *r.name; o += '\n';
        if (nb) {
            for (size_t j = 0; j < nb; j++) { if (j) o += ' '; o += gc[j] < 0 ? string("-0.1") : pct[gc[j]]; }
            o += '\n';
        } else o += "0.0\n";
    }
    if (extractNR) regions(r.nr_txt, r, cnt, nb, 1, minRepeat);
    if (extractR) regions(r.r_txt, r, cnt, nb, minRepeat, maxRepeat);
}

void Sect::regions(string& out, const Record& r, const uint64_t* cnt, size_t nb, uint32_t min_count, uint32_t max_count) {   // printRegions, :373-424
    if (!nb) return;
    const string& seq = *r.seq;
    const uint16_t k = input.merLen;
    const string maxcntstr = max_count > 0 ? string("-") + std::to_string(max_count) : "+";
    uint32_t index = 1, start = 0;
    bool inRegion = false;
    string ss;
    auto header = [&](uint32_t end) {
        out += '>'; out += *r.name;
        out += "___region:" + std::to_string(index++) + "_length:" + std::to_string((uint32_t)(end - start - 1)) + "_pos:" + std::to_string(start + 1) + ":" +
               std::to_string(end) + "_cov:" + std::to_string(min_count) + maxcntstr + "\n";
    };
    for (size_t j = 0; j < nb; j++) {
        const uint64_t c = cnt[j];
        if (c >= min_count && (c <= max_count || max_count == 0)) {
            if (!inRegion) { start = (uint32_t)j; inRegion = true; }
            ss += seq[j];
        } else if (inRegion) {
            const uint32_t end = (uint32_t)(j + k - 1);
            header(end);
            out += ss;
            for (size_t q = j + 1; q < end; q++) out += seq[q];
            out += '\n';
            inRegion = false;
            ss.clear();
        }
    }
    if (inRegion) {
        const uint32_t end = (uint32_t)(nb + k - 1);
        header(end);
        out += ss;
        for (size_t q = nb; q < end; q++) out += seq[q];
        out += '\n';
    }
}

void Sect::processSeqFile() {                                                                   // src/sect.cc:143-244
    PhaseTimer timer;
    cout << "Calculating kmer coverage across sequences ...";
    cout.flush();

    SeqRecordReader reader(seqFile);
    if (verbose) std::cerr << endl;
    std::ofstream count_path_stream, gc_count_path_stream, nr_path_stream, r_path_stream;
    if (!noCountStats) count_path_stream.open((outputPrefix + "-counts.cvg").c_str());
    if (outputGCStats) gc_count_path_stream.open((outputPrefix + "-counts.gc").c_str());
    if (extractNR) nr_path_stream.open((outputPrefix + "-non_repetitive.fa").c_str());
    if (extractR) r_path_stream.open((outputPrefix + "-repetitive.fa").c_str());
    std::ofstream cvg_gc_stream((outputPrefix + "-stats.tsv").c_str());
    cvg_gc_stream << "seq_name\tmedian\tmean\tgc%\tseq_length\tkmers_in_seq\tinvalid_kmers\t%_invalid\tnon_zero_kmers\t%_non_zero\t%_non_zero_corrected" << endl;

    // The reference reads 1024 records at a time (BATCH_SIZE, src/sect.hpp:66); the batch only bounds memory there and is
    // invisible in the outputs.  Here a batch is bounded by bases instead, so that one device call has enough to do.
    const size_t BATCH_BASES = (size_t)64 << 20;
    vector<string> names, seqs;
    vector<Record> recs;
    string joined;
    vector<uint64_t> counts;
    vector<size_t> offs;
    while (!reader.atEnd()) {
        if (verbose) std::cerr << "Loading Batch of sequences... ";
        names.clear(); seqs.clear();
        size_t bases = 0;
        while (!reader.atEnd() && bases < BATCH_BASES) {
            names.emplace_back(); seqs.emplace_back();
            reader.readRecord(names.back(), seqs.back());
            bases += seqs.back().size() + 1;
        }
        const size_t n = names.size();
        if (verbose) std::cerr << "Loaded " << n << " records.  Processing batch... ";

        // analyseBatch(): one device call for every window of every record of the batch
        joined.clear(); joined.reserve(bases);
        offs.assign(n, 0);
        for (size_t i = 0; i < n; i++) { offs[i] = joined.size(); joined += seqs[i]; joined += '\n'; }   // a newline can never be in a record
        if (counts.size() < joined.size()) counts.resize(joined.size());
        Engine::check(katgpu_table_profile_host(input.hash, joined.data(), joined.size(), input.canonical ? 1 : 0, counts.data()));

        recs.assign(n, Record());
        for (size_t i = 0; i < n; i++) { recs[i].name = &names[i]; recs[i].seq = &seqs[i]; }
        const unsigned workers = std::max<unsigned>(1, std::min<unsigned>(threads, (unsigned)n));
        auto work = [&](unsigned th) { for (size_t i = th; i < n; i += workers) processSeq(recs[i], counts.data() + offs[i]); };   // processInterlaced, :477-483
        if (workers == 1) work(0);
        else {
            vector<std::thread> team;
            for (unsigned th = 0; th < workers; th++) team.emplace_back(work, th);
            for (auto& t : team) t.join();
        }

        char line[512];
        for (size_t i = 0; i < n; i++) {
            const Record& r = recs[i];
            if (!noCountStats) count_path_stream << r.cvg_txt;
            if (outputGCStats) gc_count_path_stream << r.gc_txt;
            if (extractNR) nr_path_stream << r.nr_txt;
            if (extractR) r_path_stream << r.r_txt;
            // printStatTable, :427-445: std::fixed << setprecision(5); kmers_in_seq is uint32 arithmetic and wraps for short records
            snprintf(line, sizeof line, "\t%u\t%.5f\t%.5f\t%u\t%u\t%u\t%.5f\t%u\t%.5f\t%.5f\n", r.median, r.mean, r.gc, r.length,
                     (uint32_t)(r.length - input.merLen + 1), r.invalid, r.percentInvalid, r.nonZero, r.percentNonZero, r.percentNonZeroCorrected);
            cvg_gc_stream << *r.name << line;
            if (r.mx_x < gcBins && r.mx_y < cvgBins) contamination_mx.data()[(size_t)r.mx_x * cvgBins + r.mx_y] += seqs[i].size();
        }
        if (verbose) std::cerr << "done" << endl;
    }
    cout << " done.";
    cout.flush();
}

int Sect::main(int argc, char* argv[]) {                                                        // src/sect.cc:604-741
    static const vector<OptSpec> spec = {
        {"output_prefix", 'o', true}, {"gc_bins", 'x', true}, {"cvg_bins", 'y', true}, {"cvg_logscale", 'l', false},
        {"threads", 't', true}, {"5ptrim", 0, true}, {"non_canonical", 'N', false}, {"mer_len", 'm', true}, {"hash_size", 'H', true},
        {"no_count_stats", 'n', false}, {"output_gc_stats", 'g', false}, {"extract_nr", 'E', false}, {"extract_r", 'F', false},
        {"min_repeat", 'M', true}, {"max_repeat", 'G', true}, {"dump_hash", 'd', false}, {"verbose", 'v', false}, {"help", 0, false}};
    ParsedArgs pa = parseArgs(argc, argv, spec);
    if (pa.has("help") || argc <= 1) {
        cout << "Usage: kat sect [options] <sequence_file> (<input>)+\n\nEstimates coverage levels across sequences in the provided input sequence file.\n" << endl;
        return 1;
    }
    vector<uint16_t> trim = parseTrimList(pa.get("5ptrim", "0"));
    PhaseTimer total("KAT SECT completed.\nTotal runtime: %.1fs\n\n");
    cout << "Running KAT in SECT mode" << endl << "------------------------" << endl << endl;
    string seq_file = pa.positional.empty() ? string() : pa.positional[0];                      // p.add("seq_file", 1); p.add("counts_files", -1)
    vector<string> counts_files(pa.positional.begin() + (pa.positional.empty() ? 0 : 1), pa.positional.end());
    Sect sect(counts_files, seq_file);
    sect.setOutputPrefix(pa.get("output_prefix", "kat-sect"));
    sect.setGcBins((uint16_t)std::stoul(pa.get("gc_bins", "1001")));
    sect.setCvgBins((uint16_t)std::stoul(pa.get("cvg_bins", "1001")));
    sect.setCvgLogscale(pa.has("cvg_logscale"));
    sect.setThreads((uint16_t)std::stoul(pa.get("threads", "1")));
    sect.setTrim(trim);
    sect.setCanonical(!pa.has("non_canonical"));
    sect.setMerLen((uint16_t)std::stoul(pa.get("mer_len", std::to_string(DEFAULT_MER_LEN))));
    sect.setHashSize(std::stoull(pa.get("hash_size", std::to_string(DEFAULT_HASH_SIZE))));
    sect.setNoCountStats(pa.has("no_count_stats"));
    sect.setOutputGCStats(pa.has("output_gc_stats"));
    sect.setExtractNR(pa.has("extract_nr"));
    sect.setExtractR(pa.has("extract_r"));
    sect.setMinRepeat((uint32_t)std::stoul(pa.get("min_repeat", "2")));
    sect.setMaxRepeat((uint32_t)std::stoul(pa.get("max_repeat", "0")));
    sect.setDumpHash(pa.has("dump_hash"));
    sect.setVerbose(pa.has("verbose"));
    sect.execute();         // the reference's main stops here: Sect::save() (the contamination matrix) is never called
    if (getenv("KATGPU_TESTING") && getenv("KATGPU_SECT_SAVE")) sect.save();     // test hook: exercises Sect::save() through the CLI
    return 0;
}

// ================================================================ Cold (src/cold.cc) ==============================
// Every record of the assembly profiled against the reads hash and against the assembly's own hash: two
// katgpu_table_profile_host calls per batch, one -stats.tsv row per record.

Cold::Cold(const vector<string>& reads_files, const string& asm_file) {                          // src/cold.cc:67-77
    reads.setMultipleInputs(reads_files);
    reads.index = 1;
    assembly.setSingleInput(asm_file);
    assembly.index = 1;                                                                         // sic: both inputs are "Input 1"
    outputPrefix = "kat-cold";
}

void Cold::execute() {                                                                           // src/cold.cc:80-122
    reads.validateInput();
    assembly.validateInput();
    ensureDirectoryExists(parentOfAbsolute(outputPrefix));
    // Cold never sets InputHandler::canonical (lib/include/kat/input_handler.hpp:48 defaults it to false): counted hashes
    // are non-canonical and probed as such; a loaded .jf brings its own flag.
    if (reads.mode == InputHandler::COUNT) reads.count(threads);
    else { reads.loadHeader(); reads.loadHash(); }
    if (assembly.mode == InputHandler::COUNT) assembly.count(threads);
    else { assembly.loadHeader(); assembly.loadHash(); }
    processSeqFile();
    if (dumpHashes()) {
        reads.dump(outputPrefix + "-reads_hash.jf" + std::to_string(reads.merLen), threads);
        assembly.dump(outputPrefix + "-asm_hash.jf" + std::to_string(assembly.merLen), threads);
    }
}

void Cold::processSeq(Row& r, const string& seq, const uint64_t* readsCounts, const uint64_t* asmCounts) {   // src/cold.cc:303-408
    const uint16_t k = reads.merLen;
    const uint64_t seqLength = seq.size();
    const int64_t nbCounts = (int64_t)seqLength - k + 1;
    const size_t nb = nbCounts > 0 ? (size_t)nbCounts : 0;
    uint64_t nbNonZero = 0, nbInvalid = 0;
    r.median = 0; r.mean = 0.0; r.asmCn = 0;
    if (nb) {
        uint32_t bad = 0;
        uint64_t sum = 0;
        for (size_t i = 0; i < seqLength; i++) {                     // rolling validKmer: invalid windows count for neither sum nor nonZero
            bad += !isBase(seq[i]);
            if (i >= k) bad -= !isBase(seq[i - k]);
            if (i + 1 >= k) {
                const size_t w = i + 1 - k;
                if (bad) nbInvalid++;
                else { sum += readsCounts[w]; if (readsCounts[w]) nbNonZero++; }
            }
        }
        vector<uint64_t> sorted(readsCounts, readsCounts + nb);
        std::nth_element(sorted.begin(), sorted.begin() + nb / 2, sorted.end());
        r.median = (uint32_t)(double)sorted[nb / 2];
        r.mean = (double)sum / (double)nbCounts;
        sorted.assign(asmCounts, asmCounts + nb);
        std::nth_element(sorted.begin(), sorted.begin() + nb / 2, sorted.end());
        r.asmCn = (uint32_t)(double)sorted[nb / 2];
    }
    r.length = (uint32_t)seqLength;
    r.nonZero = (uint32_t)nbNonZero;
    r.percentNonZero = nbNonZero == 0 || nbCounts <= 0 ? 0.0 : ((double)nbNonZero / (double)nbCounts) * 100.0;
    r.invalid = (uint32_t)nbInvalid;
    r.percentInvalid = nbInvalid == 0 || nbCounts <= 0 ? 0.0 : ((double)nbInvalid / (double)nbCounts) * 100.0;
    const uint64_t notInvalid = (uint64_t)nbCounts - nbInvalid;
    r.percentNonZeroCorrected = nbNonZero == 0 || notInvalid <= 0 ? 0.0 : ((double)nbNonZero / (double)notInvalid) * 100.0;
    uint64_t gs = 0, cs = 0, ns = 0;
    for (char c : seq) {
        if (c == 'G' || c == 'g') gs++;
        else if (c == 'C' || c == 'c') cs++;
        else if (c == 'N' || c == 'n') ns++;
    }
    volatile double num = (double)(gs + cs), den = (double)(seqLength - ns);
    r.gc = num / den;
}

void Cold::processSeqFile() {                                                                    // src/cold.cc:126-200
    PhaseTimer timer;
    cout << "Calculating kmer coverage across sequences ...";
    cout.flush();
    SeqRecordReader reader(assembly.pathString());
    if (verbose) std::cerr << endl;
    std::ofstream cvg_gc_stream((outputPrefix + "-stats.tsv").c_str());
    cvg_gc_stream << "seq_name\tread_median_cvg\tread_mean_cvg\tasm_cn\tgc%\tseq_length\tkmers_in_seq\tinvalid_kmers\t%_invalid\tnon_zero_kmers\t%_non_zero\t%_non_zero_corrected" << endl;

    const size_t BATCH_BASES = (size_t)64 << 20;
    vector<string> names, seqs;
    vector<Row> rows;
    string joined;
    vector<uint64_t> rcounts, acounts;
    vector<size_t> offs;
    while (!reader.atEnd()) {
        if (verbose) std::cerr << "Loading Batch of sequences... ";
        names.clear(); seqs.clear();
        size_t bases = 0;
        while (!reader.atEnd() && bases < BATCH_BASES) {
            names.emplace_back(); seqs.emplace_back();
            reader.readRecord(names.back(), seqs.back());
            bases += seqs.back().size() + 1;
        }
        const size_t n = names.size();
        if (verbose) std::cerr << "Loaded " << n << " records.  Processing batch... ";
        joined.clear(); joined.reserve(bases);
        offs.assign(n, 0);
        for (size_t i = 0; i < n; i++) { offs[i] = joined.size(); joined += seqs[i]; joined += '\n'; }
        if (rcounts.size() < joined.size()) { rcounts.resize(joined.size()); acounts.resize(joined.size()); }
        Engine::check(katgpu_table_profile_host(reads.hash, joined.data(), joined.size(), reads.canonical ? 1 : 0, rcounts.data()));
        Engine::check(katgpu_table_profile_host(assembly.hash, joined.data(), joined.size(), assembly.canonical ? 1 : 0, acounts.data()));

        rows.assign(n, Row());
        const unsigned workers = std::max<unsigned>(1, std::min<unsigned>(threads, (unsigned)n));
        auto work = [&](unsigned th) { for (size_t i = th; i < n; i += workers) processSeq(rows[i], seqs[i], rcounts.data() + offs[i], acounts.data() + offs[i]); };
        if (workers == 1) work(0);
        else {
            vector<std::thread> team;
            for (unsigned th = 0; th < workers; th++) team.emplace_back(work, th);
            for (auto& t : team) t.join();
        }
        char line[512];
        for (size_t i = 0; i < n; i++) {                                                        // printStatTable, src/cold.cc:254-271
            const Row& r = rows[i];
            snprintf(line, sizeof line, "\t%u\t%.5f\t%u\t%.5f\t%u\t%u\t%u\t%.5f\t%u\t%.5f\t%.5f\n", r.median, r.mean, r.asmCn, r.gc, r.length,
                     (uint32_t)(r.length - assembly.merLen + 1), r.invalid, r.percentInvalid, r.nonZero, r.percentNonZero, r.percentNonZeroCorrected);
            cvg_gc_stream << names[i] << line;
        }
        if (verbose) std::cerr << "done" << endl;
    }
    cout << " done.";
    cout.flush();
}

int Cold::main(int argc, char* argv[]) {                                                         // src/cold.cc:436-546
    static const vector<OptSpec> spec = {
        {"output_prefix", 'o', true}, {"gc_bins", 'x', true}, {"cvg_bins", 'y', true}, {"threads", 't', true}, {"5ptrim", 0, true},
        {"mer_len", 'm', true}, {"hash_size", 'H', true}, {"dump_hashes", 'd', false}, {"disable_hash_grow", 'g', false},
        {"output_type", 'p', true}, {"verbose", 'v', false}, {"help", 0, false}};
    ParsedArgs pa = parseArgs(argc, argv, spec);
    if (pa.has("help") || argc <= 1) {
        cout << "Usage: kat cold [options] <assembly> (<reads>)+\n\nCalculates median read k-mer coverage, assembly k-mer coverage and GC% across each sequence in the provided assembly.\n" << endl;
        return 1;
    }
    vector<uint16_t> trim = parseTrimList(pa.get("5ptrim", "0"));
    PhaseTimer total("KAT CoLD completed.\nTotal runtime: %.1fs\n\n");
    cout << "Running KAT in Cold mode" << endl << "------------------------" << endl << endl;
    string asm_file = pa.positional.empty() ? string() : pa.positional[0];                      // p.add("asm_file", 1); p.add("reads_files", -1)
    vector<string> reads_files(pa.positional.begin() + (pa.positional.empty() ? 0 : 1), pa.positional.end());
    Cold cold(reads_files, asm_file);
    cold.setOutputPrefix(pa.get("output_prefix", "kat-cold"));
    cold.setGcBins((uint16_t)std::stoul(pa.get("gc_bins", "1001")));
    cold.setCvgBins((uint16_t)std::stoul(pa.get("cvg_bins", "1001")));
    cold.setThreads((uint16_t)std::stoul(pa.get("threads", "1")));
    cold.setReadsTrim(trim);
    cold.setMerLen((uint8_t)std::stoul(pa.get("mer_len", std::to_string(DEFAULT_MER_LEN))));   // uint8_t setter, src/cold.hpp:160
    cold.setHashSize(std::stoull(pa.get("hash_size", std::to_string(DEFAULT_HASH_SIZE))));
    cold.setDumpHashes(pa.has("dump_hashes"));
    cold.setVerbose(pa.has("verbose"));
    // --disable_hash_grow is parsed and never applied in the reference's main (src/cold.cc:525-535)
    cold.execute();         // cold.plot() needs the Python plotting package: out of scope
    return 0;
}

}  // namespace kat
