// kg_ingest.hpp -- host-side sequence ingest for libkatgpu: FASTA/FASTQ (plain or gzip) -> base stream.
//
// Replaces Jellyfish's stream_manager + mer_overlap_sequence_parser (deps/jellyfish-2.2.0/include/jellyfish/
// stream_manager.hpp:115-145, mer_overlap_sequence_parser.hpp:132-289) with a block-streaming state machine: the
// reference pulls characters through std::istream one line at a time under a cooperative thread pool; here raw
// blocks are scanned with memchr and whole lines are appended to the output, which the caller ships to the GPU.
// Semantics kept: format by first byte ('>' / '@'), header lines skipped, newlines removed, every other byte passed
// through (so IUPAC codes, '-', '\r' break k-mers downstream), records separated by one 'N', FASTQ qualities
// skipped by LENGTH (a quality line may start with '@'), 5' trim per record.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace kg {

uint64_t file_size_or_zero(const char* path);

// The record state machine, separated from the file it reads so that ranges of a file can be parsed independently
// (parse_file_parallel) with the same code as the streaming parser.
struct ParseState {
    enum Type { NONE, FASTA, FASTQ };
    enum State { HEADER, TRIM_SKIPNL, TRIM_IGNORE, LOOP_CHECK, FORCED_SKIPNL, SEQ_LINE, SEQ_SKIPNL, PLUS_LINE,
                 QUAL_SKIPNL, QUAL_IGNORE, QUAL_DONE_SKIPNL };
    Type type = NONE;
    State st = HEADER;
    uint32_t trim5p = 0;
    uint64_t trim_left = 0;
    uint64_t seq_len = 0;                  // sequence bytes of the current FASTQ record
    uint64_t read_len = 0, quals = 0;      // skip_quals bookkeeping
    uint64_t want = 0, got = 0;

    // Feed one raw block through the machine, appending base-stream bytes to out.  *bad_fastq: "Invalid fastq sequence".
    void consume(const uint8_t* d, size_t n, std::vector<uint8_t>& out, bool* bad_fastq);
    bool begin(uint8_t first_byte);        // dispatch on the first byte of the file; false: "Unsupported format"
    bool end_ok() const;                   // end of file: is the last record complete?
    // Will a '>' ('@' for FASTQ) as the next byte start a record?  (After a sequence line the machine sits in SEQ_SKIPNL until
    // it sees a byte that is not a newline, and then looks at it exactly as LOOP_CHECK does.)
    bool at_record_boundary() const { return type == FASTA ? (st == LOOP_CHECK || st == SEQ_SKIPNL) : st == QUAL_DONE_SKIPNL; }
private:
    void after_header();
};

// First GUESSED record start at file offset >= from, looking only at buf (= file bytes [buf_off, buf_off + len)): "\n>" for FASTA; "\n@", a
// line, then "\n+" for FASTQ.  -1: none is certain inside the buffer.  A guess: whoever uses it checks it (the thread team against the
// machine's state at the end of the piece before, the device scan against the structure of every record of the chunk).
int64_t find_record_start(ParseState::Type type, const uint8_t* buf, int64_t buf_off, int64_t len, int64_t from);

// Whole PLAIN four-line FASTQ records in [p, p + n) -- '@' line, one sequence line, '+' line, one quality line of the sequence's
// length, every line ended by '\n' -- stripped to what the counter reads: each record's sequence bytes (verbatim: IUPAC codes, '\r'
// and the like break k-mers downstream, as in the reference) followed by one 'N', the base stream's record separator
// (mer_overlap_sequence_parser.hpp:202,234).  out must hold n / 2 + 1 bytes.  Returns false -- and nothing of `out` is to be used --
// when the bytes are not exactly that: multi-line records, a quality line of another length (the reference skips qualities by
// LENGTH, parser.hpp:274-289: only the state machine above knows what such a file means), a record without bases, a missing final
// newline, a stray byte.
// The reader threads of kg_scan.hip run it on record-aligned pieces of large FASTQ files, so that only the bases cross PCIe.
bool strip_fastq_records(const uint8_t* p, size_t n, uint8_t* out, size_t* out_n);

class SeqFileParser {
public:
    SeqFileParser();
    ~SeqFileParser();
    SeqFileParser(const SeqFileParser&) = delete;
    SeqFileParser& operator=(const SeqFileParser&) = delete;

    // returns a katgpu_status (0 ok); *err gets the message.  raw_bytes: how much of the (inflated) file one next() call reads
    int open(const char* path, uint32_t trim5p, std::string* err, size_t raw_bytes = (size_t)16 << 20);
    // next piece of the base stream; *n == 0 means end of file
    int next(const uint8_t** p, size_t* n, std::string* err);

private:
    void* gz_ = nullptr;
    std::string path_;
    ParseState ps_;
    bool eof_ = false;
    std::vector<uint8_t> raw_, out_;
};

// Multi-threaded front end for large plain (not gzip) regular files: the file is cut at record starts, the pieces are parsed
// by a thread team with the state machine above and handed to `sink` in file order.  A cut is only a guess ("\n>" for
// FASTA; "\n@", a line, then "\n+" for FASTQ): a piece is accepted only if the piece before it -- parsed from a state known to
// be right -- ends in the record-boundary state exactly where the piece begins; otherwise the rest of the file goes through
// the machine serially from that known state.  The output is therefore byte-identical to the streaming parser's whatever the
// file looks like.  Returns a katgpu_status, or -1 when the file does not qualify (the caller then streams it).
int parse_file_parallel(const char* path, uint32_t trim5p, const std::function<int(const uint8_t*, size_t)>& sink, std::string* err);
bool team_applies(const char* path, uint32_t trim5p);      // would parse_file_parallel take this file?

// BGZF files (bgzip: a gzip file made of independent members of <= 64 KB each, their compressed size in a 'BC' extra field):
// the members are inflated by a thread team, a window of them at a time, straight into one buffer at offsets known from their
// ISIZE trailers, and the inflated bytes go through the same state machine as the streaming parser -- so the output is
// byte-identical to it, 5' trim included.  A member that is not BGZF (a plain gzip file appended, say) hands the rest of the
// file to zlib from that offset; bytes that are not gzip at all end the input, as they do for zlib.  One ordinary gzip stream
// cannot be cut this way and stays on the streaming path.  Returns a katgpu_status, or -1 when the file is not BGZF.
int parse_bgzf_parallel(const char* path, uint32_t trim5p, const std::function<int(const uint8_t*, size_t)>& sink, std::string* err);
bool bgzf_applies(const char* path);

// ONE ordinary gzip stream (what `gzip`, `pigz` and every sequencer's pipeline write) inflated by a thread team: kg_pgzip.cpp.  The
// compressed file is cut into chunks; each is entered at a deflate block start found by search and decoded without its 32 KiB of
// history (unknown bytes travel as markers), the chunks are stitched where a decoder ends exactly on the next one's entry, the
// markers resolved in order, and the inflated chunks go through the state machine from guessed record starts, as parse_file_parallel's
// pieces do.  Byte-identical to the streaming parser for any input (every guess is checked); CRC-32 / ISIZE of every member verified.
// Returns a katgpu_status, or -1 when the file is not taken (not gzip, BGZF, small, 5' trim).
int parse_gz_parallel(const char* path, uint32_t trim5p, const std::function<int(const uint8_t*, size_t)>& sink, std::string* err);
bool pgz_applies(const char* path, uint32_t trim5p);

// One input group (InputHandler::count's file list) -> the base stream the counter consumes, handed to `sink` piece by piece.
// Files never join (mer_overlap_sequence_parser.hpp:151-155), so the stream is a sequence of file pieces with an 'N' wherever
// the source changes.  Large plain files and BGZF files go through their thread teams, one file after the other.  Runs of files that have to
// stream (gzip, 5' trim, small) are read CONCURRENTLY, one reader thread per file (inflate is ~0.3 GB/s per stream, and paired
// libraries come as two or more .gz files): the sink then sees the files' blocks interleaved, and every time the source
// switches back to a file the stream carries 'N' followed by that file's previous k-1 bytes, so that each k-mer window of each
// file appears in the stream exactly once.  The k-mer MULTISET of the stream is therefore that of the files read one by one
// (tests/test_ingest_parser.py checks exactly this); the order is not, and no consumer here depends on it.
// Errors: the one from the lowest-numbered bad file, as reading the files in order would report.
// Returns a katgpu_status; a nonzero value returned by `sink` is passed through with *err left empty.
int stream_group(const char* const* paths, size_t n_paths, const uint16_t* trim5p, uint32_t k,
                 const std::function<int(const uint8_t*, size_t)>& sink, std::string* err);

}  // namespace kg
