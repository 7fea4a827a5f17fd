"""A second, independent statement of what kat hist / kat gcp / kat comp compute, written from the reference's user documentation
(doc/source/using.rst: HIST "the number of distinct k-mers having a given frequency ... the last bucket behaves as a catchall";
GCP "for each GC count and K-mer coverage level, the number of distinct K-mers", Rows = k; COMP "distinct k-mer counts for the
frequency in each input file represented by the row and column index") and the file headers (Columns / Rows), NOT from
oracle/koracle.c.  Input: the (k-mer, count) multiset of a table as two numpy arrays -- nothing else of either engine.

Used to cross-check the three reducers whose bodies (Histogram::binSlice, Gcp::analyseSlice, Comp::compareSlice) the oracle only
restates: if the oracle and this agree, and the HIP kernels agree with both, the restatement is at least not an artefact of
reading one source file one way."""
import numpy as np


def hist(counts, low=1, high=10000):
    """Rows labelled low-ish .. high+1 as kat hist prints them (default: 1 .. 10001); out[i] = distinct k-mers with frequency
    label_i, the last row also takes everything above."""
    first = low - 1 if low > 1 else 1                    # the first label kat hist prints
    labels = high + 1 - first + 1
    c = np.asarray(counts, dtype=np.uint64)
    idx = np.clip(c.astype(np.int64) - first, 0, labels - 1)       # frequencies below the first label fall into it, above the last into the catch-all
    return np.bincount(idx, minlength=labels).astype(np.uint64)


def gc_count(keys, k):
    """#G + #C of each packed k-mer (2 bits per base, A=0 C=1 G=2 T=3)."""
    x = np.asarray(keys, dtype=np.uint64)
    g = np.zeros(x.shape, np.int64)
    for i in range(k):
        b = (x >> np.uint64(2 * i)) & np.uint64(3)
        g += (b == 1) | (b == 2)
    return g


def gc_count_popcount(keys, k):
    """The same number for arrays of 10^9 k-mers (the loop above makes k passes): a base is G or C exactly when its two bits differ,
    so the count is the number of set bits of (x ^ x >> 1) at the even positions below 2k.  tests check it against gc_count."""
    x = np.asarray(keys, dtype=np.uint64)
    even = np.uint64(sum(1 << (2 * i) for i in range(k)))
    return np.bitwise_count((x ^ (x >> np.uint64(1))) & even).astype(np.int64)


def gcp(keys, counts, k, bins=1000, gc=gc_count):
    """k rows (GC count 0 .. k-1: '# Rows:<k>'), bins + 1 columns (frequency 0 .. bins, the last a catch-all)."""
    g = gc(keys, k)
    c = np.minimum(np.asarray(counts, dtype=np.uint64), np.uint64(bins)).astype(np.int64)
    keep = g < k                                          # the matrix has k rows: a k-mer made of G and C only has no row
    flat = np.bincount(g[keep] * (bins + 1) + c[keep], minlength=k * (bins + 1))
    return flat.reshape(k, bins + 1).astype(np.uint64)


def comp_matrix(keys1, counts1, keys2, counts2, bins1=1001, bins2=1001):
    """matrix[f1][f2] = distinct k-mers seen f1 times in input 1 and f2 times in input 2 (0 = absent; both inputs canonical or
    both not; scale 1; the last row / column catch everything above)."""
    k1, k2 = np.asarray(keys1, np.uint64), np.asarray(keys2, np.uint64)
    c1, c2 = np.asarray(counts1, np.uint64), np.asarray(counts2, np.uint64)
    o1, o2 = np.argsort(k1, kind="stable"), np.argsort(k2, kind="stable")
    k1, c1, k2, c2 = k1[o1], c1[o1], k2[o2], c2[o2]
    pos = np.searchsorted(k2, k1)                         # where each input-1 k-mer sits (or would sit) in input 2
    pos_c = np.minimum(pos, max(k2.size - 1, 0))
    hit = (k2[pos_c] == k1) if k2.size else np.zeros(k1.size, bool)
    f2_of_1 = np.where(hit, c2[pos_c] if k2.size else 0, 0).astype(np.uint64)
    seen2 = np.zeros(k2.size, bool)
    seen2[pos_c[hit]] = True
    r1 = np.minimum(c1, np.uint64(bins1 - 1)).astype(np.int64)
    r2 = np.minimum(f2_of_1, np.uint64(bins2 - 1)).astype(np.int64)
    mx = np.bincount(r1 * bins2 + r2, minlength=bins1 * bins2)
    only2 = np.minimum(c2[~seen2], np.uint64(bins2 - 1)).astype(np.int64)
    mx[:bins2] += np.bincount(only2, minlength=bins2)     # row 0: absent from input 1
    return mx.reshape(bins1, bins2).astype(np.uint64)
