// kat_main.cc -- `katgpu hist|gcp|comp|sect|cold ...`: the dispatcher for the KAT modes on the path, with KAT's exit codes
// (src/kat.cc:178-305: option errors 1, KAT/boost exceptions 4, std::exception 5, const char* 6, anything else 7).
#include "kat_host.hpp"

#include <cstring>
#include <iostream>

static void usage() {
    std::cout << "The K-mer Analysis Toolkit, MI355X engine (katgpu): hist | gcp | comp | sect | cold\n"
                 "Usage: katgpu <mode> [options] <inputs>   (same options as `kat <mode>`; see INTEGRATION.md)\n";
}

int main(int argc, char* argv[]) {
    int rc = 0;
    try {
        if (argc < 2 || !strcmp(argv[1], "--help") || !strcmp(argv[1], "-h")) { usage(); return 1; }
        const std::string mode = argv[1];
        // a katgpu-only switch, valid in every mode: read FASTA inputs that carry a 5' trim exactly as the reference's parser does
        // (the trim re-applied at each 4096-byte buffer fill) instead of once per record
        int kept = 2;
        for (int i = 2; i < argc; ++i) {
            if (!strcmp(argv[i], "--jellyfish_5ptrim_compat")) katgpu_ingest_jf_5ptrim_compat(1);
            else argv[kept++] = argv[i];
        }
        argc = kept;
        if (mode == "hist") rc = kat::Histogram::main(argc - 1, argv + 1);
        else if (mode == "gcp") rc = kat::Gcp::main(argc - 1, argv + 1);
        else if (mode == "comp") rc = kat::Comp::main(argc - 1, argv + 1);
        else if (mode == "sect") rc = kat::Sect::main(argc - 1, argv + 1);
        else if (mode == "cold") rc = kat::Cold::main(argc - 1, argv + 1);
        else throw kat::OptionError("Could not recognise mode string: " + mode + " (this build carries hist, gcp, comp, sect and cold)");
    } catch (kat::OptionError& e) {
        std::cerr << "Error: Parsing Command Line: " << e.what() << std::endl;
        rc = 1;
    } catch (kat::KatException& e) {
        std::cerr << e.what() << std::endl;
        rc = 4;
    } catch (std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        rc = 5;
    } catch (const char* msg) {
        std::cerr << "Error: " << msg << std::endl;
        rc = 6;
    } catch (...) {
        std::cerr << "Error: Exception of unknown type!" << std::endl;
        rc = 7;
    }
    kat::Engine::shutdown();
    return rc;
}
