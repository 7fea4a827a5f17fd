// kg_ingest.cpp -- see kg_ingest.hpp.  Pure host code (no HIP), compiled into libkatgpu.so.
#include "kg_ingest.hpp"

#include "../../include/katgpu.h"

#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace kg {

uint64_t file_size_or_zero(const char* path) {
    struct stat st;
    return (stat(path, &st) == 0 && S_ISREG(st.st_mode)) ? (uint64_t)st.st_size : 0;
}

SeqFileParser::SeqFileParser() = default;
SeqFileParser::~SeqFileParser() {
    if (gz_) gzclose((gzFile)gz_);
}

int SeqFileParser::open(const char* path, uint32_t trim5p, std::string* err) {
    path_ = path;
    trim5p_ = trim5p;
    // the reference opens every input through a gzip-aware stream (stream_manager.hpp:133-145); zlib passes plain files through
    gz_ = gzopen(path, "rb");
    if (!gz_) {
        *err = "Could not find input file at: " + path_ + "; please check the path and try again.";   // input_handler.cc:119-122
        return KATGPU_ERR_IO;
    }
    gzbuffer((gzFile)gz_, 1 << 20);
    raw_.resize((size_t)16 << 20);
    out_.reserve(raw_.size() + 16);
    return KATGPU_OK;
}

void SeqFileParser::after_header() {
    if (trim5p_ > 0) st_ = TRIM_SKIPNL;          // read_sequence: skip_newlines(); is.ignore(trim5p)   (parser.hpp:250-253)
    else st_ = LOOP_CHECK;
}

// Feed one raw block through the record state machine, appending base-stream bytes to out_.
void SeqFileParser::consume(const uint8_t* d, size_t n, bool* bad_fastq) {
    const uint8_t stop = type_ == FASTA ? '>' : '+';
    size_t i = 0;
    while (i < n) {
        switch (st_) {
        case HEADER:
        case PLUS_LINE: {                                   // ignore_line (parser.hpp:264-266)
            const uint8_t* nl = (const uint8_t*)memchr(d + i, '\n', n - i);
            if (!nl) { i = n; break; }
            i = (size_t)(nl - d) + 1;
            if (st_ == HEADER) after_header();
            else {                                           // skip_quals(seq_len + trim5p)   (parser.hpp:237,274-289)
                read_len_ = seq_len_ + trim5p_; quals_ = 0;
                st_ = read_len_ ? QUAL_SKIPNL : QUAL_DONE_SKIPNL;
            }
            break;
        }
        case TRIM_SKIPNL:
            if (d[i] == '\n') { ++i; break; }
            trim_left_ = trim5p_; st_ = TRIM_IGNORE;
            break;
        case TRIM_IGNORE: {
            size_t take = (size_t)std::min<uint64_t>(trim_left_, n - i);
            i += take; trim_left_ -= take;
            if (!trim_left_) st_ = LOOP_CHECK;
            break;
        }
        case LOOP_CHECK:                                     // "while(... && is.peek() != stop)"   (parser.hpp:254)
            if (d[i] == stop) {
                if (type_ == FASTA) { out_.push_back('N'); st_ = HEADER; }     // 'N' between records (parser.hpp:202)
                else st_ = PLUS_LINE;
            } else if (d[i] == '\n') st_ = FORCED_SKIPNL;    // blank line right after a header: the next line is read unconditionally
            else st_ = SEQ_LINE;
            break;
        case FORCED_SKIPNL:
            if (d[i] == '\n') { ++i; break; }
            st_ = SEQ_LINE;
            break;
        case SEQ_LINE: {                                     // is.get(): the rest of the line, verbatim   (parser.hpp:257)
            const uint8_t* nl = (const uint8_t*)memchr(d + i, '\n', n - i);
            size_t e = nl ? (size_t)(nl - d) : n;
            out_.insert(out_.end(), d + i, d + e);
            seq_len_ += e - i;
            i = e;
            if (nl) st_ = SEQ_SKIPNL;
            break;
        }
        case SEQ_SKIPNL:
            if (d[i] == '\n') { ++i; break; }
            st_ = LOOP_CHECK;
            break;
        case QUAL_SKIPNL:
            if (d[i] == '\n') { ++i; break; }
            want_ = read_len_ - quals_ + 1; got_ = 0; st_ = QUAL_IGNORE;
            break;
        case QUAL_IGNORE: {                                  // is.ignore(read_len - quals + 1, '\n')   (parser.hpp:280)
            size_t lim = (size_t)std::min<uint64_t>(want_ - got_, n - i);
            const uint8_t* nl = (const uint8_t*)memchr(d + i, '\n', lim);
            size_t take = nl ? (size_t)(nl - (d + i)) + 1 : lim;
            i += take; got_ += take;
            if (nl || got_ == want_) {
                quals_ += got_; ++read_len_;                 // "if(is) ++read_len"
                st_ = quals_ < read_len_ ? QUAL_SKIPNL : QUAL_DONE_SKIPNL;
            }
            break;
        }
        case QUAL_DONE_SKIPNL:
            if (d[i] == '\n') { ++i; break; }
            if (d[i] == '@') { out_.push_back('N'); seq_len_ = 0; st_ = HEADER; }   // parser.hpp:285-286,238-242
            else { *bad_fastq = true; return; }
            break;
        }
    }
}

int SeqFileParser::next(const uint8_t** p, size_t* n, std::string* err) {
    *p = nullptr; *n = 0;
    out_.clear();
    while (!eof_ && out_.empty()) {
        int r = gzread((gzFile)gz_, raw_.data(), (unsigned)raw_.size());
        if (r < 0) { *err = "read error on " + path_; return KATGPU_ERR_IO; }
        if (r == 0) {
            eof_ = true;
            if (type_ == FASTQ) {
                // Truncated record.  (Deviation, documented in DESIGN.md: a last quality line of exactly the right length
                // but without '\n' is accepted; the reference throws there and its pool swallows the exception.)
                bool ok = true;
                if (st_ == QUAL_IGNORE) ok = quals_ + got_ == read_len_;
                else if (st_ == QUAL_SKIPNL) ok = false;
                else if (st_ == PLUS_LINE) ok = seq_len_ + trim5p_ == 0;
                if (!ok) { *err = "Invalid fastq sequence"; return KATGPU_ERR_FASTQ; }
            }
            break;
        }
        size_t off = 0;
        if (type_ == NONE) {                                  // open_next_file: dispatch on the first byte (parser.hpp:171-185)
            if (raw_[0] == '>') type_ = FASTA;
            else if (raw_[0] == '@') type_ = FASTQ;
            else { *err = "Unsupported format"; return KATGPU_ERR_FORMAT; }
            st_ = HEADER; seq_len_ = 0;
        }
        bool bad = false;
        consume(raw_.data() + off, (size_t)r - off, &bad);
        if (bad) { *err = "Invalid fastq sequence"; return KATGPU_ERR_FASTQ; }
    }
    *p = out_.data(); *n = out_.size();
    return KATGPU_OK;
}

}  // namespace kg

extern "C" int katgpu_parse_file(const char* path, uint32_t trim5p, uint8_t** bases, size_t* n, const char** err_msg) {
    static thread_local std::string last;
    if (!path || !bases || !n) return KATGPU_ERR_INVALID_ARG;
    *bases = nullptr; *n = 0;
    kg::SeqFileParser parser;
    std::vector<uint8_t> all;
    int rc = parser.open(path, trim5p, &last);
    while (!rc) {
        const uint8_t* p; size_t got;
        rc = parser.next(&p, &got, &last);
        if (rc || !got) break;
        all.insert(all.end(), p, p + got);
    }
    if (rc) { if (err_msg) *err_msg = last.c_str(); return rc; }
    *bases = (uint8_t*)malloc(all.size() ? all.size() : 1);
    if (!*bases) return KATGPU_ERR_NOMEM;
    memcpy(*bases, all.data(), all.size());
    *n = all.size();
    return KATGPU_OK;
}

extern "C" void katgpu_free_host(void* p) { free(p); }
