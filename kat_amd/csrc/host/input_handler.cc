// input_handler.cc -- mirror of KAT's InputHandler (lib/src/input_handler.cc) + the file-type sniffing of
// JellyfishHelper (lib/src/jellyfish_helper.cc:258-307).  count() is the drop-in boundary: where the reference builds a
// jellyfish HashCounter and runs countSeqFile (input_handler.cc:180-202), this calls katgpu_count.
#include "kat_host.hpp"

#include <glob.h>
#include <strings.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

namespace kat {

// ---- engine singleton ----
static katgpu_ctx* g_ctx = nullptr;

int Engine::rank_ = 0, Engine::world_ = 1;
bool Engine::dist_ = false;
std::string Engine::id_file_;
static katgpu_comm* g_comm = nullptr;

katgpu_ctx* Engine::ctx() {
    if (!g_ctx) {
        // --gpus N: rank r takes device r (mod the devices there are: ranks may share one -- the exchange then goes through /dev/shm)
        int dev = -1;
        if (dist_ && world_ > 1) {
            // how many devices THIS process sees (after the fork: the HIP runtime's count, which honours HIP_ / ROCR_VISIBLE_DEVICES);
            // KATGPU_VISIBLE_DEVICES overrides it (fewer devices than there are: ranks share)
            const char* nd = getenv("KATGPU_VISIBLE_DEVICES");
            const int n = std::max(1, nd && atoi(nd) > 0 ? atoi(nd) : katgpu_device_count());
            dev = rank_ % n;
        }
        const double t0 = timing_now_ms();
        int rc = katgpu_init(dev, &g_ctx);
        timing_line("device_init", timing_now_ms() - t0);
        if (rc) throw std::runtime_error("katgpu_init failed (status " + std::to_string(rc) + "): no gfx950 device; this build has no CPU path");
        // --gpus N: the ranks meet NOW, right after the fork, while all of them are known to be alive -- not at the first exchange, which
        // a rank reaches when it has counted its share: whole files (gzip, FASTA) are dealt rank by rank, and ranks finishing minutes
        // apart would run into the rendezvous' time-outs although nothing is wrong
        if (dist_ && world_ > 1) comm();
    }
    return g_ctx;
}

katgpu_comm* Engine::comm() {
    if (g_comm || !dist_) return g_comm;
    unsigned char id[KATGPU_COMM_ID_BYTES];
    if (rank_ == 0) {
        check(katgpu_comm_unique_id(id));
        const std::string tmp = id_file_ + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) throw std::runtime_error("cannot write " + tmp);
        fclose(f);
        if (rename(tmp.c_str(), id_file_.c_str()) != 0) throw std::runtime_error("cannot publish " + id_file_);
    } else {
        FILE* f = nullptr;
        for (int tries = 0; !(f = fopen(id_file_.c_str(), "rb")); ++tries) {
            if (tries > 120000) throw std::runtime_error("rank 0 never published the communicator id (" + id_file_ + ")");
            usleep(1000);
        }
        const size_t got = fread(id, 1, sizeof id, f);
        fclose(f);
        if (got != sizeof id) throw std::runtime_error("short communicator id in " + id_file_);
    }
    check(katgpu_comm_init(ctx(), rank_, world_, id, &g_comm));
    if (speaker() && world_ > 1) {
        const char* note = katgpu_comm_transport_note(g_comm);
        std::cout << "Multi-GPU: " << world_ << " ranks, transport " << katgpu_comm_transport(g_comm) << (note && *note ? std::string(" (") + note + ")" : std::string()) << "\n";
    }
    return g_comm;
}

static std::vector<katgpu_table*> g_pending;                               // the tables of exchanges begun and not yet finished, oldest first (at most two)
static void finish_oldest() { katgpu_table* t = g_pending.front(); g_pending.erase(g_pending.begin()); Engine::check(katgpu_exchange_finish(Engine::comm(), t)); }
void Engine::finishPending() { while (!g_pending.empty()) finish_oldest(); }
bool Engine::exchangesUnderWay() { return !g_pending.empty(); }
void Engine::exchange(katgpu_table* t) { if (dist_) { finishPending(); check(katgpu_exchange_merge(comm(), t)); } }
void Engine::exchangeBegin(katgpu_table* t) { if (dist_) { while (g_pending.size() >= 2) finish_oldest(); check(katgpu_exchange_begin(comm(), t)); g_pending.push_back(t); } }
void Engine::barrier() { if (dist_) { finishPending(); check(katgpu_comm_barrier(comm())); } }
void Engine::allreduce(uint64_t* buf, size_t n) { if (dist_) { finishPending(); check(katgpu_allreduce_u64(comm(), buf, n)); } }

void Engine::shutdown() {
    if (g_comm) { katgpu_comm_free(g_comm); g_comm = nullptr; }
    if (g_ctx) { katgpu_shutdown(g_ctx); g_ctx = nullptr; }
}

void Engine::check(int status) {
    if (status == KATGPU_OK) return;
    std::string msg = g_ctx ? katgpu_last_error(g_ctx) : "katgpu error";
    switch (status) {
    case KATGPU_ERR_IO: throw InputFileException(msg);                       // boost-derived in the reference -> exit code 4
    case KATGPU_ERR_MISMATCH: throw JellyfishException(msg);
    case KATGPU_ERR_FORMAT:                                                  // std::runtime_error in the reference -> exit code 5
    case KATGPU_ERR_FASTQ:
    case KATGPU_ERR_TABLE_FULL: throw std::runtime_error(msg);
    default: throw std::runtime_error("katgpu status " + std::to_string(status) + ": " + msg);
    }
}

// ---- InputHandler ----
InputHandler::~InputHandler() {
    if (hash) katgpu_table_free(hash);      // the reference releases its shared_ptr<HashCounter> here
}

void InputHandler::setMultipleInputs(const std::vector<std::string>& inputs) {
    input = inputs;
    trim5p.assign(inputs.size(), 0);
}

void InputHandler::set5pTrim(const std::vector<uint16_t>& trim_list) {      // lib/src/input_handler.cc:51-72
    if (trim_list.empty()) trim5p.assign(input.size(), 0);
    else if (trim_list.size() == 1 && input.size() > 1) trim5p.assign(input.size(), trim_list[0]);
    else if (trim_list.size() == input.size()) trim5p = trim_list;
    else throw InputFileException("Inconsistent number of inputs and trimming settings.  Please establish your inputs before trying to set trimming vector.  Also ensure you have the same number of input files to trimming settings.");
}

bool InputHandler::isPipe(const std::string& p) { return p.rfind("/proc", 0) == 0 || p.rfind("/dev", 0) == 0; }

static std::string extension(const std::string& p) {
    size_t slash = p.find_last_of('/');
    std::string leaf = slash == std::string::npos ? p : p.substr(slash + 1);
    size_t dot = leaf.find_last_of('.');
    if (dot == std::string::npos || leaf == "." || leaf == "..") return "";
    return leaf.substr(dot);
}

bool InputHandler::isSequenceFile(const std::string& filename) {            // lib/src/jellyfish_helper.cc:270-307
    if (isPipe(filename)) return true;
    std::string ext = extension(filename);
    if (strcasecmp(ext.c_str(), ".gz") == 0) ext = extension(filename.substr(0, filename.find_last_of('.')));
    static const char* seq_exts[] = {".fastq", ".fq", ".fasta", ".fa", ".fna", ".fas", ".scafSeq"};
    for (const char* e : seq_exts) if (strcasecmp(ext.c_str(), e) == 0) return true;
    char ch = 0;
    std::fstream fin(filename, std::fstream::in);
    fin >> ch;
    return ch == '>' || ch == '@';
}

void InputHandler::validateInput() {                                        // lib/src/input_handler.cc:97-137
    if (input.size() != trim5p.size()) throw InputFileException("Inconsistent number of inputs and trimming settings.");
    for (const auto& p : input) {
        struct stat st;
        if (!isPipe(p) && stat(p.c_str(), &st) != 0)
            throw InputFileException("Could not find input file at: " + p + "; please check the path and try again.");
        mode = isSequenceFile(p) ? COUNT : LOAD;       // `start` is never cleared in the reference: the last file decides
    }
}

std::string InputHandler::pathString() const {                              // lib/src/input_handler.cc:160-169
    std::string s;
    for (const auto& p : input) s += (isPipe(p) ? std::string("<pipe>") : p) + " ";
    while (!s.empty() && isspace((unsigned char)s.back())) s.pop_back();
    return s;
}

std::string InputHandler::fileName() const {                                // lib/src/input_handler.cc:171-178
    std::string s;
    for (const auto& p : input) {
        size_t slash = p.find_last_of('/');
        s += (slash == std::string::npos ? p : p.substr(slash + 1)) + " ";
    }
    while (!s.empty() && isspace((unsigned char)s.back())) s.pop_back();
    return s;
}

// `like` (comp only): the hash this one will be compared with; the new table adopts its region grid so that the
// comparison can join region against region on the device (katgpu_table_create_like).
// `more_to_count` (comp): another input is counted right after this one -- this table's merge across GPUs travels meanwhile (Engine::exchangeBegin).
void InputHandler::count(uint16_t threads, const katgpu_table* like, bool more_to_count) {      // lib/src/input_handler.cc:180-202
    (void)threads;          // -t sized the reference's std::thread team; the GPU engine owns its own parallelism
    auto t0 = std::chrono::steady_clock::now();
    std::cout << "Input " << index << " is a sequence file.  Counting kmers for input " << index << " (" << pathString() << ") ...";
    std::cout.flush();
    std::vector<const char*> paths;
    for (const auto& p : input) paths.push_back(p.c_str());
    if (Engine::dist()) {
        // one process per GPU: every rank counts its share of the group into a table of the SAME size hint (hence the same region
        // grid), then the tables are made one by owner, in place
        if (like) Engine::check(katgpu_table_create_like(Engine::ctx(), like, merLen, canonical ? 1 : 0, hashSize, disableHashGrow ? 1 : 0, &hash));
        else Engine::check(katgpu_table_create(Engine::ctx(), merLen, canonical ? 1 : 0, hashSize, disableHashGrow ? 1 : 0, &hash));
        Engine::check(katgpu_count_files_sharded(hash, paths.data(), paths.size(), trim5p.data(), Engine::rank(), Engine::world()));
        if (more_to_count || Engine::exchangesUnderWay()) Engine::exchangeBegin(hash); else Engine::exchange(hash);     // (comp's last input: its records travel while the one before is applied)
    } else if (like) {
        Engine::check(katgpu_table_create_like(Engine::ctx(), like, merLen, canonical ? 1 : 0, hashSize, disableHashGrow ? 1 : 0, &hash));
        Engine::check(katgpu_count_files(hash, paths.data(), paths.size(), trim5p.data()));
    } else
        Engine::check(katgpu_count(Engine::ctx(), paths.data(), paths.size(), merLen, canonical ? 1 : 0, trim5p.data(), hashSize,
                                   disableHashGrow ? 1 : 0, &hash));
    // hash_counter::double_size (JF/include/jellyfish/hash_counter.hpp:204-244) says this on stdout at every doubling, once per
    // counting thread; here once per growth step of the device table (which may more than double)
    for (uint32_t i = katgpu_table_regrows(hash); i > 0; --i)
        std::cout << "\nWarning: Specified hash size insufficent - attempting to double hash size... success!\n";
    std::cout << " done.";
    std::cout.flush();
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    timing_line("count", s * 1e3, input.empty() ? "" : input[0].c_str());
    char buf[64]; snprintf(buf, sizeof buf, "  Time taken: %.1fs\n\n", s);   // auto_cpu_timer(1, "  Time taken: %ws\n\n")
    std::cout << buf;
}

void InputHandler::loadHash() {                                             // lib/src/input_handler.cc:204-219
    auto t0 = std::chrono::steady_clock::now();
    std::cout << "Loading hashes into memory...";
    std::cout.flush();
    int rc = katgpu_jf_load(Engine::ctx(), input[0].c_str(), &hash);
    if (rc) throw JellyfishException(katgpu_jf_last_error());
    canonical = katgpu_table_canonical(hash) != 0;                          // hashLoader->getCanonical() / getMerLen()
    merLen = (uint16_t)katgpu_table_k(hash);
    if (Engine::dist()) {                                                   // every rank has read the file: rank 0's copy is the run's, shared out by owner
        if (!Engine::speaker()) Engine::check(katgpu_table_clear(hash));
        Engine::exchange(hash);
    }
    std::cout << " done.";
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    char buf[64]; snprintf(buf, sizeof buf, "  Time taken: %.1fs\n\n", s);
    std::cout << buf;
}

void InputHandler::validateMerLen(uint16_t expected) {                      // lib/src/input_handler.cc:145-158
    if (mode == LOAD && hash && katgpu_table_k(hash) != expected)
        throw JellyfishException("Cannot process hashes that were created with different K-mer lengths.  Expected: " + std::to_string(expected) +
                                 ".  Key length was " + std::to_string(katgpu_table_k(hash)) + " for : " + input[0]);
}

// --gpus N: every rank holds the k-mers it owns.  The ranks are processes of one node (kat_main.cc forks them): the others leave their
// records in a file beside the output, rank 0 puts them together with its own and writes the one sorted .jf the reference writes.
static int dump_gathered(katgpu_table* hash, const std::string& outputPath) {
    const int rank = Engine::rank(), world = Engine::world();
    const bool wide = katgpu_table_k(hash) > 32;
    size_t n = 0;
    int rc = wide ? katgpu_table_export_wide(hash, nullptr, nullptr, nullptr, 0, &n) : katgpu_table_export(hash, nullptr, nullptr, 0, &n);
    if (rc) return rc;
    std::vector<uint64_t> hi(wide ? std::max<size_t>(n, 1) : 0), lo(std::max<size_t>(n, 1)), counts(std::max<size_t>(n, 1));
    if (n) rc = wide ? katgpu_table_export_wide(hash, hi.data(), lo.data(), counts.data(), n, &n) : katgpu_table_export(hash, lo.data(), counts.data(), n, &n);
    if (rc) return rc;
    lo.resize(n); counts.resize(n);
    if (wide) hi.resize(n);
    auto part = [&](int r) { return outputPath + ".rank" + std::to_string(r) + ".part"; };
    int io_bad = 0;
    if (rank != 0) {
        FILE* f = fopen(part(rank).c_str(), "wb");
        const uint64_t n64 = n;
        io_bad = !f || fwrite(&n64, 8, 1, f) != 1 || (wide && n && fwrite(hi.data(), 8, n, f) != n) || (n && fwrite(lo.data(), 8, n, f) != n) || (n && fwrite(counts.data(), 8, n, f) != n);
        if (f && fclose(f) != 0) io_bad = 1;
    }
    Engine::barrier();                                                      // every part is on disk
    if (rank == 0) {
        for (int r = 1; r < world && !io_bad; ++r) {
            FILE* f = fopen(part(r).c_str(), "rb");
            uint64_t m = 0;
            if (!f || fread(&m, 8, 1, f) != 1) { io_bad = 1; if (f) fclose(f); break; }
            const size_t at = n;
            if (wide) hi.resize(at + m);
            lo.resize(at + m); counts.resize(at + m);
            io_bad = (wide && m && fread(hi.data() + at, 8, m, f) != m) || (m && fread(lo.data() + at, 8, m, f) != m) || (m && fread(counts.data() + at, 8, m, f) != m);
            fclose(f);
            n = at + m;
        }
        for (int r = 1; r < world; ++r) unlink(part(r).c_str());
        if (!io_bad) rc = wide ? katgpu_jf_write_records_wide(outputPath.c_str(), katgpu_table_k(hash), katgpu_table_canonical(hash), hi.data(), lo.data(), counts.data(), n)
                               : katgpu_jf_write_records(outputPath.c_str(), katgpu_table_k(hash), katgpu_table_canonical(hash), lo.data(), counts.data(), n);
    }
    uint64_t bad = (uint64_t)(io_bad || rc);
    Engine::allreduce(&bad, 1);                                             // (and nobody leaves before rank 0 has read the parts)
    if (bad) throw FileSystemException("Could not gather the ranks' k-mers for " + outputPath);
    return KATGPU_OK;
}

void InputHandler::dump(const std::string& outputPath, uint16_t threads) {  // lib/src/input_handler.cc:221-243
    (void)threads;
    struct stat st;
    if (Engine::speaker() && lstat(outputPath.c_str(), &st) == 0) unlink(outputPath.c_str());
    if (mode == COUNT) {
        auto t0 = std::chrono::steady_clock::now();
        std::cout << "Dumping hash to " << outputPath << " ...";
        std::cout.flush();
        int rc = Engine::dist() && Engine::world() > 1 ? dump_gathered(hash, outputPath) : katgpu_jf_dump(hash, outputPath.c_str());
        if (rc) throw JellyfishException(*katgpu_jf_last_error() ? katgpu_jf_last_error() : katgpu_last_error(Engine::ctx()));
        std::cout << " done.";
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        char buf[64]; snprintf(buf, sizeof buf, "  Time taken: %.1fs\n\n", s);
        std::cout << buf;
    } else if (Engine::speaker() && symlink(getSingleInput().c_str(), outputPath.c_str()) != 0) {
        throw FileSystemException("Could not create symlink " + outputPath);
    }
}

std::shared_ptr<std::vector<std::string>> InputHandler::globFiles(const std::string& in) {       // :245-255
    std::vector<std::string> v;
    std::stringstream ss(in);
    std::string tok;
    size_t start = 0;
    while (true) {                                          // boost::split on ' ' keeps empty tokens
        size_t sp = in.find(' ', start);
        v.push_back(in.substr(start, sp == std::string::npos ? std::string::npos : sp - start));
        if (sp == std::string::npos) break;
        start = sp + 1;
    }
    return globFiles(v);
}

std::shared_ptr<std::vector<std::string>> InputHandler::globFiles(const std::vector<std::string>& in) {   // :263-316
    if (in.empty()) throw InputFileException("No input provided for this input group");
    glob_t gb;
    memset(&gb, 0, sizeof gb);
    int i = 0;
    for (const auto& g : in) {
        int flags = GLOB_TILDE | GLOB_NOCHECK | GLOB_BRACE;
        if (i > 0) flags |= GLOB_APPEND;
        int ret = glob(g.c_str(), flags, nullptr, &gb);
        if (ret != 0) {
            throw InputFileException(std::string("Problem globbing input pattern: ") + g + ". Non-zero return code.  Error type: " +
                                     (ret == GLOB_ABORTED ? "filesystem problem" : ret == GLOB_NOMATCH ? "no match of pattern" :
                                      ret == GLOB_NOSPACE ? "no dynamic memory" : "unknown problem"));
        }
        ++i;
    }
    auto out = std::make_shared<std::vector<std::string>>();
    for (size_t j = 0; j < gb.gl_pathc; ++j) out->push_back(gb.gl_pathv[j]);
    if (gb.gl_pathc > 0) globfree(&gb);
    if (out->empty()) out->push_back(in[0]);
    return out;
}

// ---- filesystem helpers ----
std::string parentOfAbsolute(const std::string& prefix) {
    std::string abs = prefix;
    if (abs.empty() || abs[0] != '/') {
        char cwd[4096];
        if (!getcwd(cwd, sizeof cwd)) throw FileSystemException("getcwd failed");
        abs = std::string(cwd) + "/" + prefix;
    }
    size_t slash = abs.find_last_of('/');
    return slash == 0 ? "/" : abs.substr(0, slash);
}

void ensureDirectoryExists(const std::string& dir) {                        // lib/include/kat/kat_fs.hpp:226-238
    struct stat st;
    if (stat(dir.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) return;
    std::string acc;
    std::stringstream ss(dir);
    std::string part;
    while (std::getline(ss, part, '/')) {
        acc += part + "/";
        if (part.empty()) continue;
        if (stat(acc.c_str(), &st) != 0) mkdir(acc.c_str(), 0777);
    }
    if (stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) throw FileSystemException("Could not create output directory: " + dir);
}

}  // namespace kat
