"""The placement hash of one-word tables (kat_amd/csrc/kg_device.hpp "placement"): the counterpart of Jellyfish's invertible hash +
remainder storage (JF/include/jellyfish/large_hash_array.hpp:169-171).  The partitioned counter stores only the remainder of a
k-mer's hash in its level-2 items, so everything rests on (k-mer) -> (digit 1, digit 2, remainder) being one to one and on the
inverse being the inverse.  Host edition of the same functions the kernels run, against an exact-integer model written here."""
import numpy as np
import pytest

from kat_amd.binding import place_keys

M64 = (1 << 64) - 1
M32 = 0xFFFFFFFF
G1, G2, G3 = 0x9E3779B1, 0x85EBCA6B, 0xC2B2AE35


def mix(v, c):
    """fold to 32 bits, xor-shift, one 32-bit multiply by an odd constant (place_mix)"""
    lo, hi = v & M32, (v >> 32) & M32
    x = lo ^ (((hi << 19) | (hi >> 13)) & M32)
    x ^= x >> 15
    return (x * c) & M32


def model(key, k, p1, l2):
    """exact integers: key = H : L (L = the low n1 bits); d1 = (H + g1(L)) mod p1; L = H2 : L2 (L2 = the low rb bits);
    d2 = H2 ^ g2(L2); rem = L2"""
    n = 2 * k
    hb1 = p1.bit_length() - 1
    n1 = max(n - hb1, 0)
    l2e = min(l2, n1)
    rb = n1 - l2e
    L, H = key & ((1 << n1) - 1), key >> n1
    assert H < p1
    d1 = (H + ((mix(L, G1) * p1) >> 32)) % p1
    L2, H2 = L & ((1 << rb) - 1), L >> rb
    d2 = H2 ^ ((mix(L2, G2) >> (32 - l2e)) if l2e else 0)
    return d1, d2, L2, rb


def model_offset(rem, S):
    return (mix(rem, G3) * S) >> 32


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
    d1, d2, rem, back, rb, off = place_keys(k, 584, 10, can, region_slots=8192)
    assert rb == 35 and (back == can).all()
    b = np.bincount(d1, minlength=584)
    mean = can.size / 584
    assert abs(b - mean).max() < 6 * mean ** 0.5
    sub = np.bincount(d2, minlength=1024)
    mean2 = can.size / 1024
    assert abs(sub - mean2).max() < 6 * mean2 ** 0.5
    assert all(int(off[i]) == model_offset(int(rem[i]), 8192) for i in range(0, can.size, 997))
    o = np.bincount(off.astype(np.int64), minlength=8192)
    mo = off.size / 8192
    assert 0.8 < o.var() / mo < 1.25
