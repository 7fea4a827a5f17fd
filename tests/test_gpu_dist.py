"""The device side of the multi-GPU exchange on one GPU, call by call (include/katgpu.h "region-ordered exchange": katgpu_table_extract_* /
_clear / _merge_regions, kg_exchange.hip): the records are grouped by owner, ordered by region and complete; applying every part's runs
to the emptied table restores it bit for bit; regions too small for what arrives take the direct path.  (The whole exchange between
processes -- kg_comm.hip over /dev/shm, over RCCL, over the RCCL stand-in -- is tests/test_gpu_comm.py.)"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _records(t, n_parts):
    """(sizes, region-count matrix, keys, counts32, big) of a table's extraction, downloaded."""
    eng = t.engine
    R = t.geometry().n_regions
    cnt = eng.alloc(4 * n_parts * R)
    sizes = t.extract_sizes(n_parts, cnt.ptr)
    total = int(sizes.sum())
    dk, dc = eng.alloc(8 * max(total, 1)), eng.alloc(4 * max(total, 1))
    big = t.extract(n_parts, cnt.ptr, dk.ptr, dc.ptr)
    return sizes, cnt, dk, dc, big, total


@pytest.mark.parametrize("k,canonical,n_parts", [(27, True, 3), (32, False, 8), (15, True, 1)])
def test_extract_clear_merge_roundtrip(engine, ko, k, canonical, n_parts):
    """katgpu_table_extract_* / _clear / _merge_regions on one GPU: the records are grouped by owner, ordered by region and
    complete; applying every part's runs to the emptied table restores it bit for bit."""
    from kat_amd import dist as kdist
    from kat_amd import synth
    g = synth.genome(300000, seed=5)
    stream = np.concatenate([synth.reads(g, 0, 20000, seed=2), np.frombuffer(b"N" + b"T" * 600 + b"N", np.uint8)])
    t = engine.table(k, canonical, size_hint=1 << 21).count_bases(stream)
    t.merge_host(np.array([77, 78], np.uint64), np.array([(1 << 32) + 9, 0xFFFFFFFF], np.uint64))     # one above 32 bits, one just inside
    o = ko.Table(k, canonical).count_bases(stream)
    o.add(77, (1 << 32) + 9)
    o.add(78, 0xFFFFFFFF)
    want_k, want_c = o.dump_sorted()
    geo = t.geometry()
    R = geo.n_regions
    sizes, cnt, dk, dc, (bk, bc), total = _records(t, n_parts)
    keys, c32 = dk.download(np.uint64, total), dc.download(np.uint32, total)
    m = cnt.download(np.uint32, n_parts * R).reshape(n_parts, R).astype(np.int64)
    assert np.array_equal(m.sum(1), sizes.astype(np.int64))
    base = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    for p in range(n_parts):                                               # grouped by owner
        kk = keys[base[p]:base[p + 1]]
        assert (kdist.owner_of(kk, k, n_parts) == p).all() if n_parts > 1 else True
    ones = np.uint64(0xFFFFFFFFFFFFFFFF)
    full = {int(a): int(b) for a, b in zip(keys, c32)}
    for a, b in zip(bk, bc):
        assert int(a) == int(ones) or full[int(a)] == 0                    # the in-band record of a big count carries 0
        full[int(a)] = int(b)
    assert 77 in [int(x) for x in bk] and 78 not in [int(x) for x in bk]
    got = sorted(full.items())
    assert [a for a, _ in got] == [int(x) for x in want_k] and [b for _, b in got] == [int(x) for x in want_c]
    # restore: every part is one source ordered by this table's regions
    t.clear()
    assert t.stats()["distinct"] == 0
    srcs = [(dk.ptr + 8 * int(base[p]), dc.ptr + 4 * int(base[p]), cnt.ptr + 4 * p * R, int(sizes[p]), geo.p1, geo.p2) for p in range(n_parts)]
    half = R // 2
    if half:                                                               # two chunks of consecutive regions
        off = np.concatenate([np.zeros((n_parts, 1), np.int64), np.cumsum(m, axis=1)], axis=1)
        for lo, hi in ((0, half), (half, R)):
            t.merge_regions(lo, hi, [(dk.ptr + 8 * int(base[p] + off[p][lo]), dc.ptr + 4 * int(base[p] + off[p][lo]), cnt.ptr + 4 * (p * R + lo),
                                      int(off[p][hi] - off[p][lo]), geo.p1, geo.p2) for p in range(n_parts)])
    else:
        t.merge_regions(0, R, srcs)
    t.merge_host(bk, bc)
    gk, gc = t.dump_sorted()
    assert np.array_equal(gk, want_k) and np.array_equal(gc, want_c)
    assert engine.profile()["merge"]["launches"] > 0
    # a second application doubles every count (adds into occupied regions), through the direct path this time
    t.merge_regions(0, R, [(s_[0], s_[1], None, s_[3], 0, 0) for s_ in srcs])
    t.merge_host(bk, bc)
    gk, gc = t.dump_sorted()
    assert np.array_equal(gk, want_k) and np.array_equal(gc, want_c * np.uint64(2))
    for b in (cnt, dk, dc):
        b.free()
    t.free()


def test_merge_regions_overflowing_regions_take_the_direct_path(engine, ko):
    """Owner regions too small for what arrives: k_merge_apply defers them, the host makes room and inserts directly."""
    from kat_amd import synth
    k = 27
    stream = synth.reads(synth.genome(200000, seed=9), 0, 15000, seed=3)
    a = engine.table(k, True, size_hint=1 << 21).count_bases(stream)
    geo = a.geometry()
    sizes, cnt, dk, dc, big, total = _records(a, 1)
    b = engine.table(k, True, size_hint=1 << 16, like=a)                   # same grid, regions of 256 slots (the smallest a common grid is kept for)
    gb = b.geometry()
    assert (gb.p1, gb.p2) == (geo.p1, geo.p2) and gb.region_slots < geo.region_slots
    b.merge_regions(0, geo.n_regions, [(dk.ptr, dc.ptr, cnt.ptr, total, geo.p1, geo.p2)])
    want_k, want_c = ko.Table(k, True).count_bases(stream).dump_sorted()
    gk, gc = b.dump_sorted()
    assert np.array_equal(gk, want_k) and np.array_equal(gc, want_c)
    for x in (cnt, dk, dc):
        x.free()
    a.free(); b.free()



def test_packed_records_case_in_a_process_with_tiny_regions():
    """katgpu_table_extract_packed / _merge_regions_packed (9-byte records: what a slot holds of the k-mer + its count): tests/packed_records_case.py,
    run with regions of 128 slots so that small tables are packed ones -- what every table of size is."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for lazy in ("0", "1"):                                # 1: a cleared table keeps its slots as they are and the merge is their first sweep (every table of size)
        env = dict(os.environ, KATGPU_TESTING="1", KATGPU_TEST_REGION_SLOTS="128")
        if lazy == "1":
            env["KATGPU_TEST_LAZY_MIN_SLOTS"] = "1024"
        r = subprocess.run([sys.executable, os.path.join(here, "packed_records_case.py")], env=env, capture_output=True, text=True, timeout=420)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert "packed records ok" in r.stdout
