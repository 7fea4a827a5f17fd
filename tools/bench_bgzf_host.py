"""Diagnostic (host only): a BGZF FASTQ through zlib (streaming parser) and through the inflate team (kg_ingest.hpp: parse_bgzf_parallel)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kat_amd
from tests.test_ingest_parser import bgzf_bytes
rng = np.random.default_rng(0)
n = 300_000
seqs = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n, 150))
rec = np.empty((n, 313), np.uint8)
rec[:, :8] = np.frombuffer(b"@read/1 ", np.uint8); rec[:, 8] = 10; rec[:, 9:159] = seqs; rec[:, 159] = 10
rec[:, 160:162] = np.frombuffer(b"+\n", np.uint8); rec[:, 162:312] = ord("I"); rec[:, 312] = 10
data = rec.tobytes() * 2
p = '/tmp/katgpu_bgzf_bench.fq.gz'
open(p, 'wb').write(bgzf_bytes(data, rng, 65280, 65280))
print('file', os.path.getsize(p) / 1e6, 'MB compressed,', len(data) / 1e6, 'MB text')
for mode, thr in (('0', 1), ('1', 1), ('1', 4), ('1', 16), ('1', 32)):
    os.environ.update(KATGPU_BGZF=mode, KATGPU_BGZF_THREADS=str(thr))
    t = time.time(); s = kat_amd.parse_file(p); dt = time.time() - t
    print('bgzf' if mode == '1' else 'zlib stream', thr, 'threads', f'{dt:.2f}s', f'{len(data) / dt / 1e6:.0f} MB/s of text', s.size)
os.remove(p)
