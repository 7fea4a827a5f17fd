"""ctypes wrapper around oracle/libkoracle.so -- the CPU ORACLE (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (kat_amd/) never does.  See oracle/koracle.h for what the oracle is pinned against.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64p = C.POINTER(C.c_uint64)


def build():
    """Compile libkoracle.so with gcc (plain C, a second or two)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "libkoracle.so")
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, f))
                                     for f in ("koracle.c", "koracle_sect.c", "koracle_wide.c", "koracle.h")):
        build()
    L = C.CDLL(so)
    L.ko_table_new.restype = C.c_void_p
    L.ko_table_new.argtypes = [C.c_uint, C.c_int]
    L.ko_table_free.argtypes = [C.c_void_p]
    L.ko_table_k.argtypes = [C.c_void_p]
    L.ko_table_k.restype = C.c_uint
    for f in ("ko_table_distinct", "ko_table_total"):
        getattr(L, f).restype = C.c_uint64
        getattr(L, f).argtypes = [C.c_void_p]
    L.ko_table_get.restype = C.c_uint64
    L.ko_table_get.argtypes = [C.c_void_p, C.c_uint64]
    L.ko_table_add.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.ko_table_dump_sorted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ko_count_bases.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.ko_count_bases_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.ko_parse_file.argtypes = [C.c_char_p, C.c_uint, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.ko_free.argtypes = [C.c_void_p]
    L.ko_count_files.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(C.c_uint16)]
    L.ko_jf_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), u64p, C.c_char_p, C.c_size_t]
    L.ko_hist.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]
    L.ko_gcp.argtypes = [C.c_void_p, C.c_double, C.c_uint32, C.c_void_p]
    L.ko_comp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint32, C.c_uint32,
                          C.c_void_p, C.c_void_p, C.c_void_p]
    L.ko_comp_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint32, C.c_uint32,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.ko_comp3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint32, C.c_uint32,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ko_write_comp_stats3.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.ko_write_comp_extra.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.ko_wtable_new.restype = C.c_void_p
    L.ko_wtable_new.argtypes = [C.c_uint, C.c_int]
    L.ko_wtable_free.argtypes = [C.c_void_p]
    for f in ("ko_wtable_distinct", "ko_wtable_total"):
        getattr(L, f).restype = C.c_uint64
        getattr(L, f).argtypes = [C.c_void_p]
    L.ko_wtable_add.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
    L.ko_wtable_get.restype = C.c_uint64
    L.ko_wtable_get.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.ko_wtable_dump_sorted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ko_wcount_bases.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.ko_wcount_files.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(C.c_uint16)]
    L.ko_whist.argtypes = L.ko_hist.argtypes
    L.ko_wgcp.argtypes = L.ko_gcp.argtypes
    L.ko_wcomp.argtypes = L.ko_comp.argtypes
    L.ko_wcomp3.argtypes = L.ko_comp3.argtypes
    L.ko_encode.argtypes = [C.c_char_p, C.c_uint, u64p]
    L.ko_decode.argtypes = [C.c_uint64, C.c_uint, C.c_char_p]
    L.ko_revcomp.restype = C.c_uint64
    L.ko_revcomp.argtypes = [C.c_uint64, C.c_uint]
    L.ko_canonical.restype = C.c_uint64
    L.ko_canonical.argtypes = [C.c_uint64, C.c_uint]
    L.ko_distance.restype = C.c_double
    L.ko_distance.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    cpp = C.POINTER(C.c_char_p)
    L.ko_write_hist.argtypes = [C.c_char_p, C.c_uint, cpp, C.c_size_t, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]
    L.ko_write_gcp.argtypes = [C.c_char_p, C.c_uint, cpp, C.c_size_t, C.c_uint32, C.c_void_p]
    L.ko_write_comp_main.argtypes = [C.c_char_p, C.c_uint, cpp, C.c_size_t, cpp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p]
    L.ko_write_comp_stats.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.ko_write_comp_hist.argtypes = [C.c_char_p, C.c_uint, cpp, C.c_size_t, C.c_void_p, C.c_uint32]
    L.ko_profile.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.ko_wprofile.argtypes = L.ko_profile.argtypes
    L.ko_sect.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint, C.c_uint32, C.c_uint32]
    L.ko_cold.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_char_p]
    L.ko_wsect.argtypes = L.ko_sect.argtypes
    L.ko_wcold.argtypes = L.ko_cold.argtypes
    _LIB = L
    return L


def _paths(paths):
    arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
    return arr, len(paths)


ERRORS = {1: "io error", 2: "Unsupported format", 3: "Invalid fastq sequence", 4: "k unsupported", 5: "out of memory"}


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(ERRORS.get(code, "error %d" % code))
        self.code = code


def encode(s, k=None):
    k = k or len(s)
    out = C.c_uint64()
    if lib().ko_encode(s.encode(), k, C.byref(out)):
        raise ValueError("non-ACGT base in %r" % s)
    return out.value


def decode(key, k):
    buf = C.create_string_buffer(k + 1)
    lib().ko_decode(int(key), k, buf)
    return buf.value.decode()


def revcomp(key, k):
    return lib().ko_revcomp(int(key), k)


def canonical(key, k):
    return lib().ko_canonical(int(key), k)


def parse_file(path, trim5p=0):
    """FASTA/FASTQ(.gz) -> the 'N'-joined base stream the reference's parser produces (uint8 array)."""
    p = C.c_void_p()
    n = C.c_size_t()
    rc = lib().ko_parse_file(os.fsencode(path), trim5p, C.byref(p), C.byref(n))
    if rc:
        raise OracleError(rc)
    out = np.frombuffer(C.string_at(p, n.value), dtype=np.uint8).copy() if n.value else np.zeros(0, np.uint8)
    lib().ko_free(p)
    return out


class Table:
    """(k-mer -> count) multiset with the reference's reducers."""

    def __init__(self, k, canonical=True, _handle=None):
        self.h = _handle if _handle is not None else lib().ko_table_new(k, int(bool(canonical)))
        if not self.h:
            raise OracleError(4)
        self.k = k
        self.canonical = bool(canonical)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ko_table_free(self.h)
            self.h = None

    @classmethod
    def from_jf(cls, path):
        h = C.c_void_p()
        n = C.c_uint64()
        hdr = C.create_string_buffer(1 << 16)
        rc = lib().ko_jf_load(os.fsencode(path), C.byref(h), C.byref(n), hdr, len(hdr))
        if rc:
            raise OracleError(rc)
        t = cls(lib().ko_table_k(h), _handle=h.value)
        t.n_records = n.value
        t.header_json = hdr.value.decode()
        t.canonical = '"canonical":true' in t.header_json
        return t

    def count_bases(self, bases, threads=1):
        b = np.ascontiguousarray(np.frombuffer(bases, dtype=np.uint8) if isinstance(bases, (bytes, bytearray)) else bases, dtype=np.uint8)
        if threads > 1:
            lib().ko_count_bases_mt(self.h, b.ctypes.data, b.size, threads)
        else:
            lib().ko_count_bases(self.h, b.ctypes.data, b.size)
        return self

    def count_files(self, paths, trim5p=None):
        arr, n = _paths(paths)
        tr = (C.c_uint16 * n)(*trim5p) if trim5p else None
        rc = lib().ko_count_files(self.h, arr, n, tr)
        if rc:
            raise OracleError(rc)
        return self

    def add(self, key, amount=1):
        lib().ko_table_add(self.h, int(key), int(amount))

    def get(self, key):
        return lib().ko_table_get(self.h, int(key))

    @property
    def distinct(self):
        return lib().ko_table_distinct(self.h)

    @property
    def total(self):
        return lib().ko_table_total(self.h)

    def dump_sorted(self):
        n = self.distinct
        keys = np.zeros(n, np.uint64)
        counts = np.zeros(n, np.uint64)
        lib().ko_table_dump_sorted(self.h, keys.ctypes.data, counts.ctypes.data)
        return keys, counts

    # ---- reducers ----
    def hist(self, low=1, high=10000, inc=1):
        base, ceil_, nb = hist_geometry(low, high)
        out = np.zeros(nb, np.uint64)
        lib().ko_hist(self.h, base, ceil_, inc, out.ctypes.data, nb)
        return out

    def gcp(self, cvg_scale=1.0, cvg_bins=1000):
        out = np.zeros((self.k, cvg_bins + 1), np.uint64)
        lib().ko_gcp(self.h, cvg_scale, cvg_bins, out.ctypes.data)
        return out


M64 = (1 << 64) - 1


class WideTable:
    """The same multiset for k-mers of up to 64 bases (koracle_wide.c); keys are Python ints of 2k bits, dumps are (hi, lo, counts)."""

    def __init__(self, k, canonical=True):
        self.h = lib().ko_wtable_new(k, int(bool(canonical)))
        if not self.h:
            raise OracleError(4)
        self.k = k
        self.canonical = bool(canonical)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ko_wtable_free(self.h)
            self.h = None

    def count_bases(self, bases):
        b = np.ascontiguousarray(np.frombuffer(bases, dtype=np.uint8) if isinstance(bases, (bytes, bytearray)) else bases, dtype=np.uint8)
        lib().ko_wcount_bases(self.h, b.ctypes.data, b.size)
        return self

    def count_files(self, paths, trim5p=None):
        arr, n = _paths(paths)
        tr = (C.c_uint16 * n)(*trim5p) if trim5p else None
        rc = lib().ko_wcount_files(self.h, arr, n, tr)
        if rc:
            raise OracleError(rc)
        return self

    def add(self, key, amount=1):
        lib().ko_wtable_add(self.h, int(key) >> 64, int(key) & M64, int(amount))

    def get(self, key):
        return lib().ko_wtable_get(self.h, int(key) >> 64, int(key) & M64)

    @property
    def distinct(self):
        return lib().ko_wtable_distinct(self.h)

    @property
    def total(self):
        return lib().ko_wtable_total(self.h)

    def dump_sorted(self):
        n = self.distinct
        hi, lo, counts = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        lib().ko_wtable_dump_sorted(self.h, hi.ctypes.data, lo.ctypes.data, counts.ctypes.data)
        return hi, lo, counts

    def hist(self, low=1, high=10000, inc=1):
        base, ceil_, nb = hist_geometry(low, high)
        out = np.zeros(nb, np.uint64)
        lib().ko_whist(self.h, base, ceil_, inc, out.ctypes.data, nb)
        return out

    def gcp(self, cvg_scale=1.0, cvg_bins=1000):
        out = np.zeros((self.k, cvg_bins + 1), np.uint64)
        lib().ko_wgcp(self.h, cvg_scale, cvg_bins, out.ctypes.data)
        return out


def hist_geometry(low, high):
    """Histogram::calcBase / calcCeil / nb_buckets (src/histogram.hpp:172-178, src/histogram.cc:68-70)."""
    base = low - 1 if low > 1 else 1
    ceil_ = high + 1
    return base, ceil_, ceil_ + 1 - base


def comp(t1, t2, d1_scale=1.0, d2_scale=1.0, d1_bins=1001, d2_bins=1001, threads=1):
    ss = min(d1_bins, d2_bins)
    mx = np.zeros((d1_bins, d2_bins), np.uint64)
    cc = np.zeros(13, np.uint64)
    sp = np.zeros((4, ss), np.uint64)
    if isinstance(t1, WideTable):
        lib().ko_wcomp(t1.h, t2.h, int(t1.canonical), int(t2.canonical), d1_scale, d2_scale, d1_bins, d2_bins,
                       mx.ctypes.data, cc.ctypes.data, sp.ctypes.data)
    elif threads > 1:
        lib().ko_comp_mt(t1.h, t2.h, int(t1.canonical), int(t2.canonical), d1_scale, d2_scale, d1_bins, d2_bins,
                         mx.ctypes.data, cc.ctypes.data, sp.ctypes.data, threads)
    else:
        lib().ko_comp(t1.h, t2.h, int(t1.canonical), int(t2.canonical), d1_scale, d2_scale, d1_bins, d2_bins,
                      mx.ctypes.data, cc.ctypes.data, sp.ctypes.data)
    return mx, cc, sp


def distance(which, s1, s2):
    s1 = np.ascontiguousarray(s1, np.uint64)
    s2 = np.ascontiguousarray(s2, np.uint64)
    return lib().ko_distance(which, s1.ctypes.data, s2.ctypes.data, s1.size)


# ---- writers ----
def write_hist(out_path, k, paths, low, high, inc, data):
    base, _, nb = hist_geometry(low, high)
    arr, n = _paths(paths)
    d = np.ascontiguousarray(data, np.uint64)
    assert d.size == nb
    return lib().ko_write_hist(os.fsencode(out_path), k, arr, n, base, inc, d.ctypes.data, nb)


def write_gcp(out_path, k, paths, cvg_bins, mx):
    arr, n = _paths(paths)
    m = np.ascontiguousarray(mx, np.uint64)
    return lib().ko_write_gcp(os.fsencode(out_path), k, arr, n, cvg_bins, m.ctypes.data)


def write_comp(prefix, k, paths1, paths2, d1_bins, d2_bins, mx, cc, sp, hists=False):
    a1, n1 = _paths(paths1)
    a2, n2 = _paths(paths2)
    m = np.ascontiguousarray(mx, np.uint64)
    c = np.ascontiguousarray(cc, np.uint64)
    s = np.ascontiguousarray(sp, np.uint64)
    ss = min(d1_bins, d2_bins)
    L = lib()
    L.ko_write_comp_main(os.fsencode(prefix + "-main.mx"), k, a1, n1, a2, n2, d1_bins, d2_bins, m.ctypes.data)
    L.ko_write_comp_stats(os.fsencode(prefix + ".stats"), os.fsencode(paths1[0]), os.fsencode(paths2[0]), c.ctypes.data, s.ctypes.data, ss)
    if hists:
        L.ko_write_comp_hist(os.fsencode(prefix + ".1.hist"), k, a1, n1, s[0].ctypes.data, ss)
        L.ko_write_comp_hist(os.fsencode(prefix + ".2.hist"), k, a2, n2, s[1].ctypes.data, ss)


def comp3(t1, t2, t3, d1_scale=1.0, d2_scale=1.0, d1_bins=1001, d2_bins=1001):
    ss = min(d1_bins, d2_bins)
    mxs = [np.zeros((d1_bins, d2_bins), np.uint64) for _ in range(4)]
    cc = np.zeros(13, np.uint64)
    sp = np.zeros((4, ss), np.uint64)
    fn = lib().ko_wcomp3 if isinstance(t1, WideTable) else lib().ko_comp3
    fn(t1.h, t2.h, t3.h, int(t1.canonical), int(t2.canonical), int(t3.canonical), d1_scale, d2_scale, d1_bins, d2_bins,
                   mxs[0].ctypes.data, mxs[1].ctypes.data, mxs[2].ctypes.data, mxs[3].ctypes.data, cc.ctypes.data, sp.ctypes.data)
    return mxs[0], mxs[1], mxs[2], mxs[3], cc, sp


def write_comp3(prefix, k, paths1, paths2, paths3, d1_bins, d2_bins, mxs, cc, sp, hists=False):
    """All files `kat comp` writes for three inputs: -main.mx, -ends.mx, -middle.mx, -mixed.mx, .stats (+ .N.hist)."""
    write_comp(prefix, k, paths1, paths2, d1_bins, d2_bins, mxs[0], cc, sp, hists)
    L = lib()
    c = np.ascontiguousarray(cc, np.uint64)
    s = np.ascontiguousarray(sp, np.uint64)
    L.ko_write_comp_stats3(os.fsencode(prefix + ".stats"), os.fsencode(paths1[0]), os.fsencode(paths2[0]), os.fsencode(paths3[0]),
                           c.ctypes.data, s.ctypes.data, min(d1_bins, d2_bins))
    for which, name in enumerate(("-ends.mx", "-middle.mx", "-mixed.mx")):
        m = np.ascontiguousarray(mxs[which + 1], np.uint64)
        L.ko_write_comp_extra(os.fsencode(prefix + name), which, os.fsencode(paths1[0]), os.fsencode(paths2[0]), os.fsencode(paths3[0]),
                              d1_bins, d2_bins, m.ctypes.data)


def profile(table, seq, canonical=None):
    """Per-position coverage of one sequence (src/sect.cc:516-535): (counts u64[n-k+1], gc i16[n-k+1], -1 = invalid)."""
    if isinstance(seq, str):
        seq = seq.encode()
    n = max(0, len(seq) - table.k + 1)
    counts = np.zeros(n, dtype=np.uint64)
    gcs = np.zeros(n, dtype=np.int16)
    if n:
        (lib().ko_wprofile if isinstance(table, WideTable) else lib().ko_profile)(table.h, int(table.canonical if canonical is None else canonical), seq, len(seq),
                         counts.ctypes.data, gcs.ctypes.data)
    return counts, gcs


def sect(table, seq_path, prefix, canonical=None, gc_bins=1001, cvg_bins=1001, no_count_stats=False, output_gc_stats=False,
         extract_nr=False, extract_r=False, cvg_logscale=False, min_repeat=2, max_repeat=0, save=False):
    """`kat sect` end to end: writes <prefix>-counts.cvg, -stats.tsv (+ optional files; save=True adds -contamination.mx)."""
    flags = (1 if no_count_stats else 0) | (2 if output_gc_stats else 0) | (4 if extract_nr else 0) | (8 if extract_r else 0) | \
        (16 if cvg_logscale else 0) | (32 if save else 0)
    rc = (lib().ko_wsect if isinstance(table, WideTable) else lib().ko_sect)(table.h, int(table.canonical if canonical is None else canonical), os.fsencode(seq_path), os.fsencode(prefix),
                       gc_bins, cvg_bins, flags, min_repeat, max_repeat)
    if rc:
        raise OracleError(rc)


def cold(reads, assembly, asm_path, prefix, canon_reads=None, canon_asm=None):
    """`kat cold`: writes <prefix>-stats.tsv (Cold never sets InputHandler::canonical, so counted hashes are non-canonical)."""
    rc = (lib().ko_wcold if isinstance(reads, WideTable) else lib().ko_cold)(reads.h, int(reads.canonical if canon_reads is None else canon_reads),
                       assembly.h, int(assembly.canonical if canon_asm is None else canon_asm), os.fsencode(asm_path), os.fsencode(prefix))
    if rc:
        raise OracleError(rc)
