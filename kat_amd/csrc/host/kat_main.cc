// kat_main.cc -- `katgpu hist|gcp|comp|sect|cold ...`: the dispatcher for the KAT modes on the path, with KAT's exit codes
// (src/kat.cc:178-305: option errors 1, KAT/boost exceptions 4, std::exception 5, const char* 6, anything else 7).
#include "kat_host.hpp"

#include <csignal>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include <sys/wait.h>
#include <unistd.h>

static void usage() {
    std::cout << "The K-mer Analysis Toolkit, MI355X engine (katgpu): hist | gcp | comp | sect | cold\n"
                 "Usage: katgpu <mode> [options] <inputs>   (same options as `kat <mode>`; see INTEGRATION.md)\n";
}

// `--gpus N` (katgpu only, every counting mode): N processes, one per GPU, forked here -- before anything touches the device, a HIP
// context does not survive a fork -- and joined at the end; each knows its rank (Engine::setDist) and finds the others through a file
// in which rank 0 leaves the communicator's id.  The exit code is the worst of the ranks'.
static int run_mode(const std::string& mode, int argc, char* argv[]) {
    if (mode == "hist") return kat::Histogram::main(argc, argv);
    if (mode == "gcp") return kat::Gcp::main(argc, argv);
    if (mode == "comp") return kat::Comp::main(argc, argv);
    if (mode == "sect") return kat::Sect::main(argc, argv);
    if (mode == "cold") return kat::Cold::main(argc, argv);
    throw kat::OptionError("Could not recognise mode string: " + mode + " (this build carries hist, gcp, comp, sect and cold)");
}

static int guarded(const std::string& mode, int argc, char* argv[]) {
    int rc = 0;
    try {
        rc = run_mode(mode, argc, argv);
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
    // The outputs are closed; what is left is giving tens of GB of device memory back block by block, which the driver does faster --
    // and regardless -- when the process ends.  (KATGPU_ORDERLY_EXIT=1 frees everything the long way: leak checkers, tests.)
    std::cout.flush(); std::cerr.flush(); fflush(nullptr);
    if (!getenv("KATGPU_ORDERLY_EXIT") && !kat::Engine::dist()) _exit(rc);
    kat::Engine::shutdown();
    return rc;
}

int main(int argc, char* argv[]) {
    if (argc < 2 || !strcmp(argv[1], "--help") || !strcmp(argv[1], "-h")) { usage(); return 1; }
    const std::string mode = argv[1];
    // the katgpu-only switch, valid in every counting mode:
    //   --gpus N                    N processes, one per GPU (hist, gcp, comp)
    int kept = 2, gpus = 0;
    bool gpus_given = false;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "--gpus") && i + 1 < argc) { gpus = atoi(argv[++i]); gpus_given = true; }
        else if (!strncmp(argv[i], "--gpus=", 7)) { gpus = atoi(argv[i] + 7); gpus_given = true; }
        else argv[kept++] = argv[i];
    }
    argc = kept;
    if (gpus_given && (gpus < 1 || gpus > 256)) { std::cerr << "Error: Parsing Command Line: --gpus takes 1 .. 256" << std::endl; return 1; }
    if (!gpus_given) return guarded(mode, argc - 1, argv + 1);
    if (mode != "hist" && mode != "gcp" && mode != "comp") { std::cerr << "Error: Parsing Command Line: --gpus applies to hist, gcp and comp" << std::endl; return 1; }
    // A peer that is alive but wedged (its heartbeat thread still ticking) would hang the run for ever: no single wait inside a collective
    // lasts longer than this (two hours -- a rank dealt a whole-genome .gz reaches the exchange many minutes after the others), unless the
    // caller says otherwise.  The library reads it at its first wait.
    setenv("KATGPU_COMM_MAX_WAIT_S", "7200", 0);

    // the rendezvous file lives in a directory of this run's own (0700, a fresh name): nobody else can plant a file or a link there
    char id_dir[] = "/tmp/katgpu-comm-XXXXXX";
    if (!mkdtemp(id_dir)) { std::cerr << "Error: cannot create a rendezvous directory in /tmp" << std::endl; return 5; }
    const std::string id_file = std::string(id_dir) + "/id";
    // (which device a rank takes is decided in the rank, after the fork: the HIP runtime's own count -- Engine::ctx)
    std::vector<pid_t> kids;
    for (int r = 0; r < gpus; ++r) {
        const pid_t pid = fork();
        if (pid < 0) { std::cerr << "Error: fork failed" << std::endl; for (pid_t k : kids) kill(k, SIGTERM); return 5; }
        if (pid == 0) {
            kat::Engine::setDist(r, gpus, id_file);
            if (r != 0) { if (!freopen("/dev/null", "w", stdout)) {} }          // rank 0 speaks
            const int rc = guarded(mode, argc - 1, argv + 1);
            fflush(stdout);
            _exit(rc);
        }
        kids.push_back(pid);
    }
    int first_bad = 0;                                        // the run's exit code: that of the first rank that failed
    for (size_t left = kids.size(); left; --left) {
        int st = 0;
        const pid_t k = wait(&st);
        if (k < 0) { if (!first_bad) first_bad = 5; break; }
        const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        if (rc && !first_bad) {
            first_bad = rc;
            for (pid_t o : kids) if (o != k) kill(o, SIGTERM);                    // the others would wait for it for ever
        }
    }
    unlink(id_file.c_str());
    unlink((id_file + ".tmp").c_str());
    rmdir(id_dir);
    return first_bad;
}
