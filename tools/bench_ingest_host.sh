#!/bin/bash
# Diagnostic: host-only ingest scaling (no GPU work).  usage: tools/bench_ingest_host.sh [GB] [threads...]
set -e
cd "$(dirname "$0")/.."
GB=${1:-8}; shift || true
g++ -O2 -std=c++17 -I include tools/ingest_host_bench.cc kat_amd/csrc/kg_ingest.cpp -o /tmp/ingest_host_bench -lz -lpthread
python - "$GB" <<'PY'
import sys, numpy as np
rng = np.random.default_rng(1)
n_block = 200_000
seqs = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n_block, 150))
rec = np.empty((n_block, 313), np.uint8)
rec[:, :8] = np.frombuffer(b"@read/1 ", np.uint8); rec[:, 8] = 10; rec[:, 9:159] = seqs; rec[:, 159] = 10
rec[:, 160:162] = np.frombuffer(b"+\n", np.uint8); rec[:, 162:312] = ord("I"); rec[:, 312] = 10
b = rec.tobytes()
with open('/tmp/katgpu_ingest_host.fq', 'wb') as f:
    for _ in range(max(1, int(float(sys.argv[1]) * 1e9 / len(b)))): f.write(b)
PY
/tmp/ingest_host_bench /tmp/katgpu_ingest_host.fq "${@:-8 16 32 64}"
rm -f /tmp/katgpu_ingest_host.fq
