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
#include <string>
#include <vector>

namespace kg {

uint64_t file_size_or_zero(const char* path);

class SeqFileParser {
public:
    SeqFileParser();
    ~SeqFileParser();
    SeqFileParser(const SeqFileParser&) = delete;
    SeqFileParser& operator=(const SeqFileParser&) = delete;

    // returns a katgpu_status (0 ok); *err gets the message
    int open(const char* path, uint32_t trim5p, std::string* err);
    // next piece of the base stream; *n == 0 means end of file
    int next(const uint8_t** p, size_t* n, std::string* err);

private:
    enum Type { NONE, FASTA, FASTQ };
    enum State { HEADER, TRIM_SKIPNL, TRIM_IGNORE, LOOP_CHECK, FORCED_SKIPNL, SEQ_LINE, SEQ_SKIPNL, PLUS_LINE,
                 QUAL_SKIPNL, QUAL_IGNORE, QUAL_DONE_SKIPNL };
    void consume(const uint8_t* d, size_t n, bool* bad_fastq);
    void after_header();

    void* gz_ = nullptr;
    std::string path_;
    Type type_ = NONE;
    State st_ = HEADER;
    bool eof_ = false;
    uint32_t trim5p_ = 0;
    uint64_t trim_left_ = 0;
    uint64_t seq_len_ = 0;                 // sequence bytes of the current FASTQ record
    uint64_t read_len_ = 0, quals_ = 0;    // skip_quals bookkeeping
    uint64_t want_ = 0, got_ = 0;
    std::vector<uint8_t> raw_, out_;
};

}  // namespace kg
