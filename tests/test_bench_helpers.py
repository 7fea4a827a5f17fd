"""bench.py's host-side helpers that the .fastq.gz leg stands on: zlib's crc32_combine restated (Python's zlib module does not export it),
and the count of CPUs a container is really granted.  No GPU needed."""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_crc32_combine_is_zlibs():
    rng = np.random.default_rng(3)
    for la, lb in ((0, 5), (1, 1), (7, 0), (1000, 1), (65536, 4097), (123457, 1 << 20)):
        a, b = rng.integers(0, 256, la, dtype=np.uint8).tobytes(), rng.integers(0, 256, lb, dtype=np.uint8).tobytes()
        assert bench.crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b), (la, lb)
    # ... and chained over many blocks, as gzip_records_file chains them
    blocks = [rng.integers(0, 256, int(n), dtype=np.uint8).tobytes() for n in rng.integers(1, 50000, 40)]
    crc = 0
    for i, blk in enumerate(blocks):
        crc = bench.crc32_combine(crc, zlib.crc32(blk), len(blk)) if i else zlib.crc32(blk)
    assert crc == zlib.crc32(b"".join(blocks))


def test_effective_cpus_is_a_positive_count_within_the_hosts():
    n = bench.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
