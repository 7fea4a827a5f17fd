"""Run in a subprocess by tests/test_gpu_dist.py with KATGPU_TEST_REGION_SLOTS=128 (tiny regions: small tables become packed tables whose
remainders have at most 44 bits, like every table of size): katgpu_table_extract_packed / katgpu_table_merge_regions_packed against the 12-byte
records, katgpu_place_keys' remainders and the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import kat_amd  # noqa: E402
from kat_amd import binding as kb  # noqa: E402
from kat_amd import synth  # noqa: E402
from oracle import koracle as ko  # noqa: E402


def _records(t, n_parts):
    eng = t.engine
    R = t.geometry().n_regions
    cnt = eng.alloc(4 * n_parts * R)
    sizes = t.extract_sizes(n_parts, cnt.ptr)
    total = int(sizes.sum())
    dk, dc = eng.alloc(8 * max(total, 1)), eng.alloc(4 * max(total, 1))
    big = t.extract(n_parts, cnt.ptr, dk.ptr, dc.ptr)
    return sizes, cnt, dk, dc, big, total


def _records_packed(t, n_parts):
    eng = t.engine
    R = t.geometry().n_regions
    cnt = eng.alloc(4 * n_parts * R)
    sizes = t.extract_sizes(n_parts, cnt.ptr)
    total = int(sizes.sum())
    lo, hi, dc = eng.alloc(4 * max(total, 1)), eng.alloc(max(total, 1)), eng.alloc(4 * max(total, 1))
    big = t.extract_packed(n_parts, cnt.ptr, lo.ptr, hi.ptr, dc.ptr)
    return sizes, cnt, lo, hi, dc, big, total


def roundtrip(engine, k, canonical, n_parts, wide_rem=False):
    """katgpu_table_extract_packed / _merge_regions_packed: 9-byte records (what a slot holds of the k-mer + its count) say the same as the
    12-byte ones -- against katgpu_place_keys' remainders record by record -- and applied to the emptied table they restore it bit for bit;
    applied to a table that has ANOTHER grid by then, the k-mers come back from the sender's grid (the direct path)."""
    g = synth.genome(300000, seed=5)
    stream = np.concatenate([synth.reads(g, 0, 20000, seed=2), np.frombuffer(b"N" + b"T" * 600 + b"N", np.uint8)])
    t = engine.table(k, canonical, size_hint=1 << 21).count_bases(stream)
    t.merge_host(np.array([77, 78], np.uint64), np.array([(1 << 32) + 9, (1 << 31) + 5], np.uint64))
    assert t.packed_records()
    o = ko.Table(k, canonical).count_bases(stream)
    o.add(77, (1 << 32) + 9)
    o.add(78, (1 << 31) + 5)
    want_k, want_c = o.dump_sorted()
    geo = t.geometry()
    R = geo.n_regions
    # the two forms of one table's records, side by side
    sizes, cnt, dk, dc, (bk, bc), total = _records(t, n_parts)
    sizes2, cnt2, lo, hi, dc2, (bk2, bc2), total2 = _records_packed(t, n_parts)
    assert total2 == total and np.array_equal(sizes, sizes2)
    keys = dk.download(np.uint64, total)
    m = cnt.download(np.uint32, n_parts * R).reshape(n_parts, R).astype(np.int64)
    # (inside a (part, region) run the two extractions may order records differently: compare them as sets per run)
    pd1, pd2, prem, _, rb = kb.place_keys(k, geo.p1, int(geo.p2).bit_length() - 1, keys)
    assert rb <= 44 and (not wide_rem or rb > 40), rb
    xs = max(rb, 40) - 40                                                  # the count word's low bits that are the remainder's top ones
    w9 = dc2.download(np.uint32, total).astype(np.uint64)
    c12, c9 = dc.download(np.uint32, total), (w9 >> np.uint64(xs))
    rem = lo.download(np.uint32, total).astype(np.uint64) | (hi.download(np.uint8, total).astype(np.uint64) << np.uint64(32)) | ((w9 & np.uint64((1 << xs) - 1)) << np.uint64(40))
    big12, big9 = dict(zip(bk.tolist(), bc.tolist())), dict(zip(bk2.tolist(), bc2.tolist()))
    assert big12 == {77: (1 << 32) + 9} and big9 == ({77: (1 << 32) + 9, 78: (1 << 31) + 5} if xs else big12), (big12, big9)      # (a count beyond the record's 32 - xs bits travels out of band)
    pos = 0
    for p in range(n_parts):
        for r in range(R):
            n = int(m[p, r])
            if n:
                # (a record whose count travels out of band carries count 0; what it says of the k-mer then is not compared)
                a = sorted((x, c) if c and kk not in big9 else (0, 0) for x, c, kk in zip(prem[pos:pos + n].tolist(), c12[pos:pos + n].tolist(), keys[pos:pos + n].tolist()))
                b = sorted((x, c) if c else (0, 0) for x, c in zip(rem[pos:pos + n].tolist(), c9[pos:pos + n].tolist()))
                assert a == b, (p, r)
                assert (pd1[pos:pos + n].astype(np.int64) * geo.p2 + pd2[pos:pos + n] == r).all()
            pos += n
    bk, bc = bk2, bc2
    # restore from the packed form, two chunks of consecutive regions
    base = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    off = np.concatenate([np.zeros((n_parts, 1), np.int64), np.cumsum(m, axis=1)], axis=1)
    t.clear()
    half = max(R // 2, 1)
    for a0, a1 in ((0, half), (half, R)):
        if a1 > a0:
            t.merge_regions_packed(a0, a1, [(lo.ptr + 4 * int(base[p] + off[p][a0]), hi.ptr + int(base[p] + off[p][a0]), dc2.ptr + 4 * int(base[p] + off[p][a0]),
                                            cnt.ptr + 4 * (p * R + a0), int(off[p][a1] - off[p][a0]), geo.p1, geo.p2) for p in range(n_parts)])
    t.merge_host(bk, bc)
    gk, gc = t.dump_sorted()
    assert np.array_equal(gk, want_k) and np.array_equal(gc, want_c)
    # ... and into a table of another grid: the k-mers are rebuilt from the sender's
    u = engine.table(k, canonical, size_hint=1 << 22 if wide_rem else 1 << 17)       # (packed records go into packed tables: at k = 28, 29 a smaller table than the sender's is not one)
    gu = u.geometry()
    assert (gu.p1, gu.p2) != (geo.p1, geo.p2), (gu.p1, gu.p2)
    u.merge_regions_packed(0, R, [(lo.ptr + 4 * int(base[p]), hi.ptr + int(base[p]), dc2.ptr + 4 * int(base[p]), cnt.ptr + 4 * p * R, int(sizes[p]), geo.p1, geo.p2) for p in range(n_parts)])
    u.merge_host(bk, bc)
    gk, gc = u.dump_sorted()
    assert np.array_equal(gk, want_k) and np.array_equal(gc, want_c)
    for b in (cnt, cnt2, dk, dc, lo, hi, dc2):
        b.free()
    t.free(); u.free()


def too_small(engine):
    """Packed records into owner regions that have no room for them: deferred, room made (the grid changes), the k-mers rebuilt from the sender's grid."""
    k = 27
    sa = synth.reads(synth.genome(600000, seed=9), 0, 22000, seed=3)
    sb = synth.reads(synth.genome(600000, seed=10), 0, 22000, seed=4)
    a = engine.table(k, True, size_hint=1 << 21).count_bases(sa)
    geo = a.geometry()
    sizes, cnt, lo, hi, dc, big, total = _records_packed(a, 1)
    b = engine.table(k, True, size_hint=1 << 21, like=a).count_bases(sb)      # the same grid, half full already
    gb = b.geometry()
    assert (gb.p1, gb.p2, gb.region_slots) == (geo.p1, geo.p2, geo.region_slots) and b.regrows == 0, (gb.p1, gb.p2, gb.region_slots, b.regrows)
    b.merge_regions_packed(0, geo.n_regions, [(lo.ptr, hi.ptr, dc.ptr, cnt.ptr, total, geo.p1, geo.p2)])
    assert b.regrows > 0                                                      # (regions that could not take their runs: the direct path after growth)
    want_k, want_c = ko.Table(k, True).count_bases(sa).count_bases(sb).dump_sorted()
    gk, gc = b.dump_sorted()
    assert np.array_equal(gk, want_k) and np.array_equal(gc, want_c)
    for x in (cnt, lo, hi, dc):
        x.free()
    a.free(); b.free()

if __name__ == "__main__":
    eng = kat_amd.Engine(0)
    for k, canonical, n_parts, wide_rem in ((27, True, 3, False), (25, False, 8, False), (21, True, 1, False), (29, True, 4, True), (28, False, 2, True)):     # (the last two: remainders of more than 40 bits, the count word carries their top ones)
        roundtrip(eng, k, canonical, n_parts, wide_rem)
    too_small(eng)
    print("packed records ok")
