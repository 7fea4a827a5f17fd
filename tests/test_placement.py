"""The placement hash of one-word tables (kat_amd/csrc/kg_device.hpp "placement"): the counterpart of Jellyfish's invertible hash +
remainder storage (JF/include/jellyfish/large_hash_array.hpp:169-171).  The partitioned counter stores only the remainder of a
k-mer's hash in its level-2 items, so everything rests on (k-mer) -> (digit 1, digit 2, remainder) being one to one and on the
inverse being the inverse.  Host edition of the same functions the kernels run, against an exact-integer model written here."""
import numpy as np
import pytest

from kat_amd.binding import place_keys

M64 = (1 << 64) - 1
C1, C2 = 0xff51afd7ed558ccd, 0xc4ceb9fe1a85ec53


def model(key, k, p1, l2):
    """exact integers: two multiply / xor-shift stages; digit 1 = floor(top32(y1) * p1 / 2^32), r1 = y1 - (first y1 of that digit)"""
    n = 2 * k
    top32 = lambda y: (y >> (n - 32)) if n >= 32 else (y << (32 - n)) & 0xFFFFFFFF
    def base1(d):
        if d == 0:
            return 0
        tb = -((-(d << 32)) // p1)
        return tb << (n - 32) if n >= 32 else -((-tb) // (1 << (32 - n)))
    widest = max((((1 << n) if d + 1 == p1 else base1(d + 1)) - base1(d) - 1 for d in range(p1)), default=0)
    n1 = max(widest, 0).bit_length()
    rb = n1 - min(l2, n1)
    y1 = ((key ^ (key >> ((n + 1) // 2))) * C1) & ((1 << n) - 1)
    d1 = (top32(y1) * p1) >> 32
    r1 = y1 - base1(d1)
    assert 0 <= r1 < (1 << n1)
    y2 = ((r1 ^ (r1 >> ((n1 + 1) // 2))) * C2) & ((1 << n1) - 1)
    return d1, y2 >> rb, y2 & ((1 << rb) - 1), rb


CASES = [(27, 584, 10), (27, 1024, 9), (31, 777, 10), (32, 1000, 10), (32, 1, 0), (16, 5, 3), (15, 37, 6), (7, 1, 0), (5, 3, 2),
         (5, 40, 3), (3, 100, 4), (2, 10, 9), (1, 1, 0), (1, 3, 1), (20, 1023, 10), (13, 64, 6)]


@pytest.mark.parametrize("k,p1,l2", CASES)
def test_placement_is_one_to_one_and_inverts(k, p1, l2):
    rng = np.random.default_rng(k * 1000 + p1)
    n = 2 * k
    if n <= 16:
        keys = np.arange(1 << n, dtype=np.uint64)
    else:
        keys = rng.integers(0, 1 << 63, 50000, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 50000, dtype=np.uint64)
        keys &= np.uint64((1 << n) - 1) if n < 64 else np.uint64(M64)
        keys = np.unique(np.concatenate([keys, np.array([0, (1 << n) - 1 if n < 64 else M64], dtype=np.uint64)]))
    d1, d2, rem, back, rb = place_keys(k, p1, l2, keys)
    assert (back == keys).all()
    assert (d1 < p1).all() and (d2 < (1 << l2)).all()
    assert rb <= 64 and (rb == 64 or (rem < np.uint64(1 << rb)).all())
    triples = np.stack([d1.astype(np.uint64), d2.astype(np.uint64), rem], axis=1)
    assert np.unique(triples, axis=0).shape[0] == keys.size
    for i in rng.integers(0, keys.size, 200):                     # the kernels' arithmetic = the exact-integer definition
        m = model(int(keys[i]), k, p1, l2)
        assert (int(d1[i]), int(d2[i]), int(rem[i]), rb) == m


def test_placement_spreads_canonical_kmers_evenly():
    """canonical 27-mers of a random sequence over 584 x 1024 regions: region and bucket loads look Poisson (the segmented level 1
    and the one-pass level 2 size their runs from that)"""
    rng = np.random.default_rng(7)
    k, G = 27, 3_000_000
    bases = rng.integers(0, 4, G, dtype=np.uint64)
    key = np.zeros(G - k + 1, dtype=np.uint64)
    rc = np.zeros(G - k + 1, dtype=np.uint64)
    for i in range(k):
        key = (key << np.uint64(2)) | bases[i:G - k + 1 + i]
        rc |= (np.uint64(3) - bases[i:G - k + 1 + i]) << np.uint64(2 * i)
    can = np.minimum(key, rc)
    d1, d2, rem, back, rb = place_keys(k, 584, 10, can)
    assert rb == 35 and (back == can).all()
    b = np.bincount(d1, minlength=584)
    mean = can.size / 584
    assert abs(b - mean).max() < 6 * mean ** 0.5
    sub = np.bincount(d2, minlength=1024)
    mean2 = can.size / 1024
    assert abs(sub - mean2).max() < 6 * mean2 ** 0.5
    off = ((rem >> np.uint64(rb - 32)) * np.uint64(8192)) >> np.uint64(32)
    o = np.bincount(off.astype(np.int64), minlength=8192)
    mo = can.size / 8192
    assert 0.8 < o.var() / mo < 1.25
