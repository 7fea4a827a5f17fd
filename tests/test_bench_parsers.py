"""bench.py's `end_to_end.result_check` reads the files `katgpu` wrote back into numbers (parse_stats / parse_mx / parse_hist) and compares
them with the resident path's result.  Here the parsers are held against files written by the oracle's byte-exact writers
(oracle/koracle.c: ko_write_comp_main / _stats, ko_write_hist, ko_write_gcp -- the reference's formats): what went in comes back.  No GPU."""
import numpy as np

import bench
from oracle import koracle as ko


def test_comp_files_round_trip(tmp_path):
    rng = np.random.default_rng(5)
    mx = rng.integers(0, 1 << 40, size=(41, 33), dtype=np.uint64)
    mx[3, 7] = np.uint64((1 << 63) + 12345)                                  # beyond a signed 64-bit parse
    cc = np.array([37200000000, 999974000, 0, 2935250888, 999973990, 0, 5, 6, 7, 8, 9, 10, 11], np.uint64)
    sp = rng.integers(0, 1 << 30, size=(4, 33), dtype=np.uint64)
    ko.write_comp(str(tmp_path / "o"), 27, ["a b.fq", 'we"ird.fq'], ["asm.fa"], 41, 33, mx, cc, sp)
    assert bench.parse_stats((tmp_path / "o.stats").read_text()) == [int(x) for x in cc]
    got = bench.parse_mx(str(tmp_path / "o-main.mx"))
    assert got.dtype == np.uint64 and np.array_equal(got, mx)


def test_three_input_stats_carry_hash_3(tmp_path):
    mxs = [np.zeros((5, 5), np.uint64) for _ in range(4)]
    cc = np.arange(101, 114, dtype=np.uint64)
    sp = np.ones((4, 5), np.uint64)
    ko.write_comp3(str(tmp_path / "t"), 21, ["r1"], ["r2"], ["asm"], 5, 5, mxs, cc, sp)
    assert bench.parse_stats((tmp_path / "t.stats").read_text()) == [int(x) for x in cc]


def test_hist_and_gcp_files_round_trip(tmp_path):
    h = np.arange(10001, dtype=np.uint64) * np.uint64(3)
    ko.write_hist(str(tmp_path / "h"), 27, ["r.fq"], 1, 10000, 1, h)
    assert np.array_equal(bench.parse_hist(str(tmp_path / "h")), h)
    g = np.random.default_rng(1).integers(0, 1 << 50, size=(27, 1001), dtype=np.uint64)
    ko.write_gcp(str(tmp_path / "g.mx"), 27, ["r.fq"], 1000, g)
    assert np.array_equal(bench.parse_mx(str(tmp_path / "g.mx")), g)
