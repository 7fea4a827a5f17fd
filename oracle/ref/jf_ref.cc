// jf_ref.cc -- driver around the REAL Jellyfish 2.2.0 code of the reference (test infrastructure, like everything in oracle/).
//
// Compiled by oracle/Makefile (`make ref`) against the sources where they lie under /root/reference -- headers from
// deps/jellyfish-2.2.0/include, plus lib/mer_dna.cc, lib/rectangular_binary_matrix.cc and lib/jsoncpp.cpp -- into
// oracle/_ref/jf_ref.  Nothing of the reference is copied and nothing is stubbed: Jellyfish's headers include <config.h>
// only under HAVE_CONFIG_H, and the three lib files used here do not include it at all.  (hash_counter / large_hash_array
// cannot be driven this way: lib/allocators_mmap.cc, misc.cc and storage.cc include the generated config.h unconditionally.)
//
//   jf_ref kmers <k> <canonical 0|1> <file>...     every k-mer the reference's parser + mer_iterator deliver, counted in a
//                                                  std::map: "<kmer> <count>" lines in key order -- the reference's count
//                                                  semantics (mer_overlap_sequence_parser.hpp + mer_iterator.hpp)
//   jf_ref kmerst <k> <canonical> <t1,t2,..> <file>...   the same with KAT's 5' trim list (one value per file), through the
//                                                  trim5p_list constructor KAT added to the vendored parser
//   jf_ref merops <k> <kmer>...                    mer_dna: 2-bit word, reverse complement, canonical form of each k-mer
//   jf_ref jfread <file.jf>                        file_header + binary_reader: header fields, then "<kmer> <count> <pos>"
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include <jellyfish/binary_dumper.hpp>
#include <jellyfish/file_header.hpp>
#include <jellyfish/mer_dna.hpp>
#include <jellyfish/mer_iterator.hpp>
#include <jellyfish/mer_overlap_sequence_parser.hpp>
#include <jellyfish/stream_manager.hpp>

typedef std::vector<const char*> file_vector;
typedef jellyfish::stream_manager<file_vector::const_iterator> stream_manager_t;
typedef jellyfish::mer_overlap_sequence_parser<stream_manager_t> parser_t;
typedef jellyfish::mer_iterator<parser_t, jellyfish::mer_dna> iterator_t;

int main(int argc, char* argv[]) {
    if (argc < 3) return 2;
    const std::string mode = argv[1];
    if (mode == "kmers" && argc >= 5) {
        const unsigned k = atoi(argv[2]);
        const bool canonical = atoi(argv[3]) != 0;
        jellyfish::mer_dna::k(k);
        file_vector files(argv + 4, argv + argc);
        stream_manager_t streams(files.begin(), files.end(), 1);            // one file at a time, as JellyfishHelper::countSeqFile does
        parser_t parser(k, streams.nb_streams(), 3, 4096, streams);         // 4096-byte buffers: jellyfish_helper.cc / count_main.cc
        std::map<std::string, unsigned long> counts;
        for (iterator_t it(parser, canonical); it; ++it) counts[it->to_str()]++;
        for (std::map<std::string, unsigned long>::const_iterator kv = counts.begin(); kv != counts.end(); ++kv)
            std::cout << kv->first << ' ' << kv->second << '\n';
        return 0;
    }
    if (mode == "kmerst" && argc >= 6) {
        const unsigned k = atoi(argv[2]);
        const bool canonical = atoi(argv[3]) != 0;
        jellyfish::mer_dna::k(k);
        std::vector<uint16_t> trims;
        for (char* tok = strtok(argv[4], ","); tok; tok = strtok(NULL, ",")) trims.push_back((uint16_t)atoi(tok));
        file_vector files(argv + 5, argv + argc);
        stream_manager_t streams(files.begin(), files.end(), 1);
        parser_t parser(k, streams.nb_streams(), 3, 4096, streams, trims);
        std::map<std::string, unsigned long> counts;
        for (iterator_t it(parser, canonical); it; ++it) counts[it->to_str()]++;
        for (std::map<std::string, unsigned long>::const_iterator kv = counts.begin(); kv != counts.end(); ++kv)
            std::cout << kv->first << ' ' << kv->second << '\n';
        return 0;
    }
    if (mode == "merops" && argc >= 4) {
        const unsigned k = atoi(argv[2]);
        jellyfish::mer_dna::k(k);
        for (int i = 3; i < argc; ++i) {
            jellyfish::mer_dna m(argv[i]);
            jellyfish::mer_dna rc = m.get_reverse_complement(), can = m.get_canonical();
            std::cout << m.to_str() << ' ' << m.get_bits(0, 2 * k > 64 ? 64 : 2 * k) << ' ' << rc.to_str() << ' ' << can.to_str() << ' ' << (m < rc) << '\n';
        }
        return 0;
    }
    if (mode == "jfread") {
        std::ifstream is(argv[2], std::ios::binary);
        if (!is.good()) return 3;
        jellyfish::file_header header;
        header.read(is);
        jellyfish::mer_dna::k(header.key_len() / 2);
        std::cout << "format " << header.format() << "\nkey_len " << header.key_len() << "\nval_len " << header.val_len() << "\ncounter_len " << header.counter_len()
                  << "\nsize " << header.size() << "\nmax_reprobe " << header.max_reprobe() << "\ncanonical " << header.canonical() << "\noffset " << header.offset() << '\n';
        jellyfish::binary_reader<jellyfish::mer_dna, uint64_t> reader(is, &header);
        while (reader.next()) std::cout << reader.key().to_str() << ' ' << reader.val() << ' ' << reader.pos() << '\n';
        return 0;
    }
    return 2;
}
