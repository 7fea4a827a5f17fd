// seqan_ref.cc -- driver around the REAL SeqAn 2.0.0 sequence-file reader that `kat sect` / `kat cold` use (test infrastructure).
// Compiled by oracle/Makefile (`make ref`) against the header-only library where it lies (deps/seqan-library-2.0.0/include)
// into oracle/_ref/seqan_ref.  Reads a file exactly the way Sect::processSeqFile does (src/sect.cc:167-206: SeqFileIn +
// readRecords in batches of 1024 into StringSet<CharString>) and prints every record as "<len> <name>\n<len> <sequence>\n".
#include <iostream>

#include <seqan/seq_io.h>

int main(int argc, char* argv[]) {
    if (argc < 2) return 2;
    try {
        seqan::SeqFileIn reader(argv[1]);
        seqan::StringSet<seqan::CharString> names, seqs;
        while (!seqan::atEnd(reader)) {
            seqan::clear(names);
            seqan::clear(seqs);
            seqan::readRecords(names, seqs, reader, 1024);
            for (unsigned i = 0; i < seqan::length(names); ++i) {
                std::cout << seqan::length(names[i]) << ' ';
                std::cout.write(seqan::toCString(names[i]), seqan::length(names[i]));
                std::cout << '\n' << seqan::length(seqs[i]) << ' ';
                std::cout.write(seqan::toCString(seqs[i]), seqan::length(seqs[i]));
                std::cout << '\n';
            }
        }
    } catch (std::exception& e) {
        std::cout << "EXCEPTION " << e.what() << '\n';
        return 5;
    }
    return 0;
}
