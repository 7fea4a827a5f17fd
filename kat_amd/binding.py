"""ctypes binding of libkatgpu.so (include/katgpu.h) -- the only way Python reaches the HIP engine.

There is no CPU path here: if the shared library is missing, or no gfx950 device is visible, construction of
`Engine` raises.  (The CPU oracle lives under oracle/ and is imported by tests and the bench baseline only.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkatgpu.so")
if os.environ.get("KATGPU_TESTING") and os.environ.get("KATGPU_LIB_PATH"):      # same-box A/B of two builds of the library (tools/; dead without KATGPU_TESTING)
    LIB_PATH = os.environ["KATGPU_LIB_PATH"]

KERNEL_CLASSES = ("count", "regrow", "hist", "gcp", "comp_pass1", "comp_pass2", "partition", "merge", "part_l1_count", "part_l2", "part_apply", "part_l1_scatter", "profile", "scan")

STATUS = {
    0: "ok", 1: "invalid argument", 2: "io", 3: "Unsupported format", 4: "Invalid fastq sequence",
    5: "out of device memory", 6: "k unsupported", 7: "Hash full", 8: "device error", 9: "k mismatch",
}

# every symbol include/katgpu.h declares (tests check the .so exports all of them)
EXPORTS = (
    "katgpu_init", "katgpu_shutdown", "katgpu_last_error", "katgpu_version", "katgpu_sync", "katgpu_release_scratch", "katgpu_scratch_acquire",
    "katgpu_count", "katgpu_table_create", "katgpu_table_create_like", "katgpu_count_files", "katgpu_count_bases_host",
    "katgpu_count_bases_device", "katgpu_table_free", "katgpu_table_stats", "katgpu_table_k",
    "katgpu_table_canonical", "katgpu_table_get", "katgpu_table_profile_host", "katgpu_table_profile_device",
    "katgpu_table_export", "katgpu_hist", "katgpu_gcp",
    "katgpu_comp", "katgpu_comp3", "katgpu_table_partition_sizes", "katgpu_table_partition", "katgpu_table_merge_device",
    "katgpu_table_merge_host", "katgpu_table_geometry", "katgpu_table_extract_sizes", "katgpu_table_extract", "katgpu_table_clear",
    "katgpu_table_merge_device32", "katgpu_table_merge_regions", "katgpu_profile_reset", "katgpu_profile_get", "katgpu_dev_alloc",
    "katgpu_dev_free", "katgpu_dev_upload", "katgpu_dev_download", "katgpu_dev_mem_info",
    "katgpu_synth_genome_device", "katgpu_synth_reads_device", "katgpu_parse_file", "katgpu_parse_files", "katgpu_free_host", "katgpu_strip_fastq", "katgpu_inflate_file",
    "katgpu_table_get_wide", "katgpu_table_export_wide", "katgpu_table_merge_host_wide",
    "katgpu_table_partition_wide", "katgpu_table_merge_device_wide", "katgpu_table_regrows",
    "katgpu_jf_load", "katgpu_jf_dump", "katgpu_jf_write_records", "katgpu_jf_read_records", "katgpu_jf_last_error",
    "katgpu_jf_write_records_wide", "katgpu_jf_read_records_wide", "katgpu_place_keys", "katgpu_reserve", "katgpu_device_count",
    "katgpu_count_files_sharded", "katgpu_table_slot_bytes", "katgpu_comm_unique_id", "katgpu_comm_init", "katgpu_comm_free", "katgpu_comm_rank", "katgpu_comm_world", "katgpu_comm_transport",
    "katgpu_comm_transport_note", "katgpu_comm_distinct_devices", "katgpu_comm_barrier", "katgpu_exchange_merge", "katgpu_allreduce_u64", "katgpu_comm_stats",
    "katgpu_table_packed_records", "katgpu_table_extract_packed", "katgpu_table_merge_regions_packed", "katgpu_comm_wire", "katgpu_exchange_begin", "katgpu_exchange_finish",
)


class Geometry(C.Structure):
    _fields_ = [("k", C.c_uint32), ("canonical", C.c_uint32), ("n_regions", C.c_uint32), ("region_slots", C.c_uint32),
                ("p1", C.c_uint32), ("p2", C.c_uint32), ("capacity", C.c_uint64)]


class MergeSource(C.Structure):
    _fields_ = [("dev_keys", C.c_void_p), ("dev_counts", C.c_void_p), ("dev_region_counts", C.c_void_p), ("n_records", C.c_uint64),
                ("p1", C.c_uint32), ("p2", C.c_uint32)]


class MergeSourcePacked(C.Structure):
    _fields_ = [("dev_rem_lo", C.c_void_p), ("dev_rem_hi", C.c_void_p), ("dev_counts", C.c_void_p), ("dev_region_counts", C.c_void_p), ("n_records", C.c_uint64),
                ("p1", C.c_uint32), ("p2", C.c_uint32)]


BIG_CAP = 4200          # side-table entries + the all-ones key: what katgpu_table_extract can return out of band


class KatGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("katgpu status %d (%s): %s" % (code, STATUS.get(code, "?"), msg))
        self.code = code
        self.message = msg


_lib = None


def load_library():
    """dlopen libkatgpu.so and declare prototypes.  Needs the HIP runtime but not a GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "katgpu has no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t
    pp = C.POINTER(C.c_void_p)
    cpp = C.POINTER(C.c_char_p)
    pu64 = C.POINTER(C.c_uint64)
    L.katgpu_init.argtypes = [C.c_int, pp]
    L.katgpu_shutdown.argtypes = [vp]
    L.katgpu_shutdown.restype = None
    L.katgpu_last_error.argtypes = [vp]
    L.katgpu_last_error.restype = C.c_char_p
    L.katgpu_version.restype = C.c_char_p
    L.katgpu_sync.argtypes = [vp]
    L.katgpu_release_scratch.argtypes = [vp]
    L.katgpu_scratch_acquire.argtypes = [vp, sz, pp, C.POINTER(sz)]
    L.katgpu_count.argtypes = [vp, cpp, sz, u32, C.c_int, C.POINTER(C.c_uint16), u64, C.c_int, pp]
    L.katgpu_table_create.argtypes = [vp, u32, C.c_int, u64, C.c_int, pp]
    L.katgpu_table_create_like.argtypes = [vp, vp, u32, C.c_int, u64, C.c_int, pp]
    L.katgpu_count_files.argtypes = [vp, cpp, sz, C.POINTER(C.c_uint16)]
    L.katgpu_count_bases_host.argtypes = [vp, vp, sz]
    L.katgpu_count_bases_device.argtypes = [vp, vp, sz]
    L.katgpu_table_free.argtypes = [vp]
    L.katgpu_table_free.restype = None
    L.katgpu_table_stats.argtypes = [vp, pu64, pu64, pu64]
    L.katgpu_table_k.argtypes = [vp]
    L.katgpu_table_k.restype = u32
    L.katgpu_table_canonical.argtypes = [vp]
    L.katgpu_table_get.argtypes = [vp, vp, sz, C.c_int, vp]
    L.katgpu_table_profile_host.argtypes = [vp, vp, sz, C.c_int, vp]
    L.katgpu_table_profile_device.argtypes = [vp, vp, sz, C.c_int, vp]
    L.katgpu_table_export.argtypes = [vp, vp, vp, sz, C.POINTER(sz)]
    L.katgpu_hist.argtypes = [vp, u64, u64, u64, vp, sz]
    L.katgpu_gcp.argtypes = [vp, C.c_double, u32, vp]
    L.katgpu_comp.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, u32, u32, vp, vp, vp]
    L.katgpu_comp3.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, u32, u32, vp, vp, vp, vp, vp, vp]
    L.katgpu_table_partition_sizes.argtypes = [vp, u32, vp]
    L.katgpu_table_partition.argtypes = [vp, u32, vp, vp, vp]
    L.katgpu_table_merge_device.argtypes = [vp, vp, vp, sz]
    L.katgpu_table_merge_host.argtypes = [vp, vp, vp, sz]
    L.katgpu_table_geometry.argtypes = [vp, vp]
    L.katgpu_table_extract_sizes.argtypes = [vp, u32, vp, vp]
    L.katgpu_table_extract.argtypes = [vp, u32, vp, vp, vp, vp, vp, u32, C.POINTER(u32)]
    L.katgpu_table_clear.argtypes = [vp]
    L.katgpu_table_merge_device32.argtypes = [vp, vp, vp, sz]
    L.katgpu_table_merge_regions.argtypes = [vp, u32, u32, u32, vp]
    L.katgpu_table_packed_records.argtypes = [vp]
    L.katgpu_table_extract_packed.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp, u32, C.POINTER(u32)]
    L.katgpu_table_merge_regions_packed.argtypes = [vp, u32, u32, u32, vp]
    L.katgpu_profile_reset.argtypes = [vp]
    L.katgpu_profile_get.argtypes = [vp, C.c_int, pu64, C.POINTER(C.c_double), pu64]
    L.katgpu_dev_alloc.argtypes = [vp, sz, pp]
    L.katgpu_dev_free.argtypes = [vp, vp]
    L.katgpu_dev_upload.argtypes = [vp, vp, vp, sz]
    L.katgpu_dev_download.argtypes = [vp, vp, vp, sz]
    L.katgpu_dev_mem_info.argtypes = [vp, pu64, pu64]
    L.katgpu_synth_genome_device.argtypes = [vp, vp, u64, u64, u64]
    L.katgpu_synth_reads_device.argtypes = [vp, vp, u64, vp, u64, u64, u32, u32, u32, u64]
    L.katgpu_parse_file.argtypes = [C.c_char_p, u32, pp, C.POINTER(sz), cpp]
    L.katgpu_table_get_wide.argtypes = [vp, vp, vp, sz, C.c_int, vp]
    L.katgpu_table_export_wide.argtypes = [vp, vp, vp, vp, sz, C.POINTER(sz)]
    L.katgpu_table_merge_host_wide.argtypes = [vp, vp, vp, vp, sz]
    L.katgpu_table_regrows.argtypes = [vp]
    L.katgpu_table_regrows.restype = u32
    L.katgpu_table_slot_bytes.argtypes = [vp]
    L.katgpu_table_slot_bytes.restype = u32
    L.katgpu_table_partition_wide.argtypes = [vp, u32, vp, vp, vp, vp]
    L.katgpu_table_merge_device_wide.argtypes = [vp, vp, vp, vp, sz]
    L.katgpu_parse_files.argtypes = [vp, sz, vp, u32, pp, C.POINTER(sz), cpp]
    L.katgpu_free_host.argtypes = [vp]
    L.katgpu_free_host.restype = None
    L.katgpu_jf_load.argtypes = [vp, C.c_char_p, pp]
    L.katgpu_jf_dump.argtypes = [vp, C.c_char_p]
    L.katgpu_jf_write_records.argtypes = [C.c_char_p, u32, C.c_int, vp, vp, sz]
    L.katgpu_jf_read_records.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(C.c_int), pp, pp, C.POINTER(sz)]
    L.katgpu_jf_write_records_wide.argtypes = [C.c_char_p, u32, C.c_int, vp, vp, vp, sz]
    L.katgpu_jf_read_records_wide.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(C.c_int), pp, pp, pp, C.POINTER(sz)]
    L.katgpu_jf_last_error.restype = C.c_char_p
    _lib = L
    return L


def parse_file(path, trim5p=0):
    """Host-only ingest: FASTA/FASTQ(.gz) -> base stream (uint8 array).  Needs no GPU."""
    L = load_library()
    p, n, msg = C.c_void_p(), C.c_size_t(), C.c_char_p()
    rc = L.katgpu_parse_file(os.fsencode(path), trim5p, C.byref(p), C.byref(n), C.byref(msg))
    if rc:
        raise KatGpuError(rc, (msg.value or b"").decode(errors="replace"))
    out = np.frombuffer(C.string_at(p, n.value), dtype=np.uint8).copy() if n.value else np.zeros(0, np.uint8)
    L.katgpu_free_host(p)
    return out


def inflate_file(path, keep=True):
    """katgpu_inflate_file: the bytes of one gzip stream, inflated by the thread team (kg_pgzip.cpp); keep=False: their number (the bytes are
    inflated and checked, not kept).  Needs no GPU."""
    L = load_library()
    L.katgpu_inflate_file.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_char_p)]
    p, n, msg = C.c_void_p(), C.c_size_t(), C.c_char_p()
    rc = L.katgpu_inflate_file(os.fsencode(path), C.byref(p) if keep else None, C.byref(n), C.byref(msg))
    if rc:
        raise KatGpuError(rc, (msg.value or b"").decode(errors="replace"))
    if not keep:
        return n.value
    try:
        return C.string_at(p.value, n.value)
    finally:
        L.katgpu_free_host(p)


def strip_fastq(data):
    """katgpu_strip_fastq: whole plain four-line FASTQ records -> their sequence lines, each followed by 'N' (bytes), or None when the
    bytes are not exactly that."""
    L = load_library()
    L.katgpu_strip_fastq.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
    out = C.create_string_buffer(len(data) // 2 + 1)
    n = C.c_size_t()
    rc = L.katgpu_strip_fastq(bytes(data), len(data), out, C.byref(n))
    return out.raw[:n.value] if rc == 0 else None


def parse_files(paths, k, trim5p=None):
    """Host-only ingest of one input group: the base stream katgpu_count_files feeds to the counter (files that stream are
    read concurrently and interleaved; see include/katgpu.h).  Needs no GPU."""
    L = load_library()
    arr, n_paths = _cpaths(paths)
    tr = (C.c_uint16 * n_paths)(*trim5p) if trim5p else None
    p, n, msg = C.c_void_p(), C.c_size_t(), C.c_char_p()
    rc = L.katgpu_parse_files(arr, n_paths, tr, k, C.byref(p), C.byref(n), C.byref(msg))
    if rc:
        raise KatGpuError(rc, (msg.value or b"").decode(errors="replace"))
    out = np.frombuffer(C.string_at(p, n.value), dtype=np.uint8).copy() if n.value else np.zeros(0, np.uint8)
    L.katgpu_free_host(p)
    return out


def place_keys(k, p1, l2, keys, region_slots=0):
    """Host edition of the one-word tables' placement hash: (d1, d2, remainder, inverse(key), remainder bits) and, with region_slots,
    the home slots as a sixth value.  Needs no GPU."""
    L = load_library()
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    n = keys.size
    d1, d2 = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    rem, back = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    rb = C.c_uint32()
    off = np.zeros(n if region_slots else 0, np.uint32)
    L.katgpu_place_keys.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    rc = L.katgpu_place_keys(k, p1, l2, keys.ctypes.data, n, d1.ctypes.data, d2.ctypes.data, rem.ctypes.data, back.ctypes.data, C.addressof(rb),
                             region_slots, off.ctypes.data if region_slots else None)
    if rc:
        raise KatGpuError(rc, "katgpu_place_keys")
    return (d1, d2, rem, back, rb.value, off) if region_slots else (d1, d2, rem, back, rb.value)


def hist_geometry(low, high):
    """Histogram::calcBase / calcCeil / nb_buckets (KAT src/histogram.hpp:172-178, src/histogram.cc:68-70)."""
    base = low - 1 if low > 1 else 1
    ceil_ = high + 1
    return base, ceil_, ceil_ + 1 - base


def _cpaths(paths):
    return (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths]), len(paths)


def jf_read_records(path):
    """Host-only .jf reader: (k, canonical, keys, counts)."""
    L = load_library()
    k, can, n = C.c_uint32(), C.c_int(), C.c_size_t()
    pk, pc = C.c_void_p(), C.c_void_p()
    rc = L.katgpu_jf_read_records(os.fsencode(path), C.byref(k), C.byref(can), C.byref(pk), C.byref(pc), C.byref(n))
    if rc:
        raise KatGpuError(rc, L.katgpu_jf_last_error().decode(errors="replace"))
    keys = np.frombuffer(C.string_at(pk, n.value * 8), dtype=np.uint64).copy() if n.value else np.zeros(0, np.uint64)
    counts = np.frombuffer(C.string_at(pc, n.value * 8), dtype=np.uint64).copy() if n.value else np.zeros(0, np.uint64)
    L.katgpu_free_host(pk)
    L.katgpu_free_host(pc)
    return k.value, bool(can.value), keys, counts


def jf_write_records(path, k, canonical, keys, counts):
    """Host-only .jf writer (binary/sorted, 4-byte saturated counters)."""
    L = load_library()
    kk = np.ascontiguousarray(keys, np.uint64)
    cc = np.ascontiguousarray(counts, np.uint64)
    rc = L.katgpu_jf_write_records(os.fsencode(path), k, int(bool(canonical)), kk.ctypes.data, cc.ctypes.data, kk.size)
    if rc:
        raise KatGpuError(rc, L.katgpu_jf_last_error().decode(errors="replace"))


def jf_read_records_wide(path):
    """Host-only .jf reader for any k <= 63: (k, canonical, keys_hi, keys_lo, counts)."""
    L = load_library()
    k, can, n = C.c_uint32(), C.c_int(), C.c_size_t()
    ph, pk, pc = C.c_void_p(), C.c_void_p(), C.c_void_p()
    rc = L.katgpu_jf_read_records_wide(os.fsencode(path), C.byref(k), C.byref(can), C.byref(ph), C.byref(pk), C.byref(pc), C.byref(n))
    if rc:
        raise KatGpuError(rc, L.katgpu_jf_last_error().decode(errors="replace"))
    out = [np.frombuffer(C.string_at(p, n.value * 8), dtype=np.uint64).copy() if n.value else np.zeros(0, np.uint64) for p in (ph, pk, pc)]
    for p in (ph, pk, pc):
        L.katgpu_free_host(p)
    return k.value, bool(can.value), out[0], out[1], out[2]


def jf_write_records_wide(path, k, canonical, keys_hi, keys_lo, counts):
    """Host-only .jf writer for any k <= 63 (binary/sorted, 4-byte saturated counters)."""
    L = load_library()
    hh, kk, cc = (np.ascontiguousarray(x, np.uint64) for x in (keys_hi, keys_lo, counts))
    rc = L.katgpu_jf_write_records_wide(os.fsencode(path), k, int(bool(canonical)), hh.ctypes.data, kk.ctypes.data, cc.ctypes.data, kk.size)
    if rc:
        raise KatGpuError(rc, L.katgpu_jf_last_error().decode(errors="replace"))


class ScratchView:
    """nbytes of device memory owned by the engine, exposed through the CUDA array interface (uint8)."""

    def __init__(self, ptr, nbytes):
        self.ptr = ptr
        self.nbytes = nbytes
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2, "strides": None}


class DeviceBuffer:
    """A raw HBM allocation owned by an Engine."""

    def __init__(self, engine, nbytes):
        self.engine = engine
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        engine._chk(engine.L.katgpu_dev_alloc(engine.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr, offset=0):
        a = np.ascontiguousarray(arr)
        assert offset + a.nbytes <= self.nbytes
        self.engine._chk(self.engine.L.katgpu_dev_upload(self.engine.h, self.ptr + offset, a.ctypes.data, a.nbytes))

    def download(self, dtype=np.uint8, count=None, offset=0):
        dt = np.dtype(dtype)
        n = (self.nbytes - offset) // dt.itemsize if count is None else count
        out = np.empty(n, dt)
        self.engine._chk(self.engine.L.katgpu_dev_download(self.engine.h, out.ctypes.data, self.ptr + offset, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.engine.L.katgpu_dev_free(self.engine.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One katgpu context == one HIP device (one process per GPU)."""

    def __init__(self, device=-1):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.katgpu_init(device, C.byref(h))
        if rc:
            raise KatGpuError(rc, "katgpu_init failed: no usable gfx950 device (katgpu has no CPU fallback)")
        self.h = h.value

    def close(self):
        if getattr(self, "h", None):
            self.L.katgpu_shutdown(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise KatGpuError(rc, self.L.katgpu_last_error(self.h).decode(errors="replace"))

    def sync(self):
        self._chk(self.L.katgpu_sync(self.h))

    def release_scratch(self):
        """Hand cached device memory (parked table arrays, partition arena) back to the driver."""
        self._chk(self.L.katgpu_release_scratch(self.h))

    def scratch(self, nbytes):
        """Borrow the engine's arena as raw device scratch: an object torch / cupy can wrap without copying
        (`torch.as_tensor(obj, device="cuda")` through __cuda_array_interface__)."""
        p, got = C.c_void_p(), C.c_size_t()
        self._chk(self.L.katgpu_scratch_acquire(self.h, int(nbytes), C.byref(p), C.byref(got)))
        v = ScratchView(p.value, int(nbytes))
        v.capacity = int(got.value)             # what the arena holds (>= nbytes): the caller may size itself to it
        return v

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def mem_info(self):
        f, t = C.c_uint64(), C.c_uint64()
        self._chk(self.L.katgpu_dev_mem_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    # ---- tables ----
    def table(self, k, canonical=True, size_hint=0, disable_grow=False, like=None):
        """like: a Table this one will be compared with -- it adopts that table's region grid (fast join in comp)."""
        if like is not None:
            h = C.c_void_p()
            self._chk(self.L.katgpu_table_create_like(self.h, like.h, k, int(bool(canonical)), size_hint, int(bool(disable_grow)), C.byref(h)))
            return Table(self, k, canonical, _handle=h.value)
        return Table(self, k, canonical, size_hint, disable_grow)

    def count(self, paths, k, canonical=True, trim5p=None, size_hint=0, disable_grow=False):
        """InputHandler::count for one input group."""
        arr, n = _cpaths(paths)
        tr = (C.c_uint16 * n)(*trim5p) if trim5p else None
        h = C.c_void_p()
        self._chk(self.L.katgpu_count(self.h, arr, n, k, int(bool(canonical)), tr, size_hint, int(bool(disable_grow)), C.byref(h)))
        return Table(self, k, canonical, _handle=h.value)

    def load_jf(self, path):
        """HashLoader::loadHash: a .jf file -> table (k and canonical from its header)."""
        h = C.c_void_p()
        rc = self.L.katgpu_jf_load(self.h, os.fsencode(path), C.byref(h))
        if rc:
            raise KatGpuError(rc, self.L.katgpu_jf_last_error().decode(errors="replace"))
        return Table(self, self.L.katgpu_table_k(h), bool(self.L.katgpu_table_canonical(h)), _handle=h.value)

    # ---- profiling ----
    def profile_reset(self):
        self._chk(self.L.katgpu_profile_reset(self.h))

    def profile(self):
        out = {}
        for i, name in enumerate(KERNEL_CLASSES):
            n, ms, units = C.c_uint64(), C.c_double(), C.c_uint64()
            self._chk(self.L.katgpu_profile_get(self.h, i, C.byref(n), C.byref(ms), C.byref(units)))
            out[name] = {"launches": n.value, "ms": ms.value, "units": units.value}
        return out

    # ---- synthetic workload on the device ----
    def synth_genome(self, n_bases, seed, contig_len=0):
        """n_bases of genome; with contig_len > 0 the assembly base stream ('N' after every contig)."""
        n_out = n_bases + (n_bases // contig_len if contig_len else 0)
        buf = self.alloc(n_out)
        self._chk(self.L.katgpu_synth_genome_device(self.h, buf.ptr, n_out, seed, contig_len))
        return buf

    def synth_reads(self, genome_buf, genome_len, first_read, n_reads, read_len=150, frag_len=350, err_ppm=2000, seed=1, out=None):
        nbytes = n_reads * (read_len + 1)
        buf = out if out is not None else self.alloc(nbytes)
        assert buf.nbytes >= nbytes
        self._chk(self.L.katgpu_synth_reads_device(self.h, genome_buf.ptr, genome_len, buf.ptr, first_read, n_reads,
                                                   read_len, frag_len, err_ppm, seed))
        return buf


class Table:
    """HBM-resident (k-mer -> count) table: the replacement for InputHandler::hash (a jellyfish LargeHashArray)."""

    def __init__(self, engine, k, canonical=True, size_hint=0, disable_grow=False, _handle=None):
        self.engine = engine
        self.k = k
        self.canonical = bool(canonical)
        if _handle is None:
            h = C.c_void_p()
            engine._chk(engine.L.katgpu_table_create(engine.h, k, int(self.canonical), size_hint, int(bool(disable_grow)), C.byref(h)))
            _handle = h.value
        self.h = _handle

    def free(self):
        if getattr(self, "h", None) and getattr(self.engine, "h", None):
            self.engine.L.katgpu_table_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    # ---- counting ----
    def count_files(self, paths, trim5p=None):
        arr, n = _cpaths(paths)
        tr = (C.c_uint16 * n)(*trim5p) if trim5p else None
        self.engine._chk(self.engine.L.katgpu_count_files(self.h, arr, n, tr))
        return self

    def count_files_sharded(self, paths, rank, world, trim5p=None):
        arr, n = _cpaths(paths)
        tr = (C.c_uint16 * n)(*trim5p) if trim5p else None
        self.engine._chk(self.engine.L.katgpu_count_files_sharded(self.h, arr, n, tr, int(rank), int(world)))
        return self

    def count_bases(self, bases):
        """bases: host uint8 array / bytes (copied through pinned staging) or a DeviceBuffer (counted in place)."""
        if isinstance(bases, DeviceBuffer):
            return self.count_bases_device(bases.ptr, bases.nbytes)
        b = np.frombuffer(bases, dtype=np.uint8) if isinstance(bases, (bytes, bytearray)) else np.ascontiguousarray(bases, dtype=np.uint8)
        self.engine._chk(self.engine.L.katgpu_count_bases_host(self.h, b.ctypes.data, b.size))
        return self

    def count_bases_device(self, dev_ptr, n):
        self.engine._chk(self.engine.L.katgpu_count_bases_device(self.h, dev_ptr, n))
        return self

    # ---- inspection ----
    @property
    def regrows(self):
        """How often the table had to grow: the size hint (-H) was too small."""
        return int(self.engine.L.katgpu_table_regrows(self.h))

    def stats(self, want_total=True):
        d, t, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.engine._chk(self.engine.L.katgpu_table_stats(self.h, C.byref(d), C.byref(t) if want_total else None, C.byref(c)))
        return {"distinct": d.value, "total": t.value if want_total else None, "capacity": c.value}

    def get(self, keys, canonicalise=False):
        k = np.ascontiguousarray(keys, np.uint64)
        out = np.zeros(k.size, np.uint64)
        self.engine._chk(self.engine.L.katgpu_table_get(self.h, k.ctypes.data, k.size, int(bool(canonicalise)), out.ctypes.data))
        return out

    def profile(self, seq, canonicalise=None):
        """Per-position coverage of `seq` (bytes / str / uint8 array): u64[len-k+1], 0 for windows with a non-base."""
        if isinstance(seq, str):
            seq = seq.encode()
        b = np.frombuffer(seq, np.uint8) if isinstance(seq, (bytes, bytearray)) else np.ascontiguousarray(seq, np.uint8)
        out = np.zeros(max(0, b.size - self.k + 1), np.uint64)
        canon = self.canonical if canonicalise is None else canonicalise
        self.engine._chk(self.engine.L.katgpu_table_profile_host(self.h, b.ctypes.data, b.size, int(bool(canon)), out.ctypes.data))
        return out

    def profile_device(self, dev_bases, n, dev_counts, canonicalise=None):
        """Device-resident form: `dev_bases` / `dev_counts` are DeviceBuffers or raw device addresses."""
        canon = self.canonical if canonicalise is None else canonicalise
        pb = getattr(dev_bases, "ptr", dev_bases)
        pc = getattr(dev_counts, "ptr", dev_counts)
        self.engine._chk(self.engine.L.katgpu_table_profile_device(self.h, pb, n, int(bool(canon)), pc))

    def export(self):
        n = C.c_size_t()
        self.engine._chk(self.engine.L.katgpu_table_export(self.h, None, None, 0, C.byref(n)))
        keys = np.zeros(n.value, np.uint64)
        counts = np.zeros(n.value, np.uint64)
        if n.value:
            self.engine._chk(self.engine.L.katgpu_table_export(self.h, keys.ctypes.data, counts.ctypes.data, n.value, C.byref(n)))
        return keys, counts

    def dump_sorted(self):
        if self.k > 32:
            return self.dump_sorted_wide()
        keys, counts = self.export()
        order = np.argsort(keys, kind="stable")
        return keys[order], counts[order]

    # ---- wide tables (33 <= k <= 63): a k-mer is (hi, lo), the upper and lower 64 bits of its 2k-bit word ----
    def export_wide(self):
        n = C.c_size_t()
        self.engine._chk(self.engine.L.katgpu_table_export_wide(self.h, None, None, None, 0, C.byref(n)))
        hi, lo, counts = np.zeros(n.value, np.uint64), np.zeros(n.value, np.uint64), np.zeros(n.value, np.uint64)
        if n.value:
            self.engine._chk(self.engine.L.katgpu_table_export_wide(self.h, hi.ctypes.data, lo.ctypes.data, counts.ctypes.data, n.value, C.byref(n)))
        return hi, lo, counts

    def dump_sorted_wide(self):
        hi, lo, counts = self.export_wide()
        order = np.lexsort((lo, hi))
        return hi[order], lo[order], counts[order]

    def merge_host_wide(self, hi, lo, counts):
        h, l, c = (np.ascontiguousarray(x, np.uint64) for x in (hi, lo, counts))
        assert h.size == l.size == c.size
        self.engine._chk(self.engine.L.katgpu_table_merge_host_wide(self.h, h.ctypes.data, l.ctypes.data, c.ctypes.data, h.size))

    def partition_wide(self, n_parts, offsets, dev_hi_ptr, dev_lo_ptr, dev_counts_ptr):
        off = np.ascontiguousarray(offsets, np.uint64)
        self.engine._chk(self.engine.L.katgpu_table_partition_wide(self.h, n_parts, off.ctypes.data, dev_hi_ptr, dev_lo_ptr, dev_counts_ptr))

    def merge_device_wide(self, dev_hi_ptr, dev_lo_ptr, dev_counts_ptr, n):
        self.engine._chk(self.engine.L.katgpu_table_merge_device_wide(self.h, dev_hi_ptr, dev_lo_ptr, dev_counts_ptr, n))

    def get_wide(self, hi, lo, canonicalise=False):
        h, l = np.ascontiguousarray(hi, np.uint64), np.ascontiguousarray(lo, np.uint64)
        out = np.zeros(h.size, np.uint64)
        self.engine._chk(self.engine.L.katgpu_table_get_wide(self.h, h.ctypes.data, l.ctypes.data, h.size, int(bool(canonicalise)), out.ctypes.data))
        return out

    def dump_jf(self, path):
        """InputHandler::dump: write the table as a Jellyfish binary/sorted hash."""
        rc = self.engine.L.katgpu_jf_dump(self.h, os.fsencode(path))
        if rc:
            raise KatGpuError(rc, self.engine.L.katgpu_jf_last_error().decode(errors="replace") or self.engine.L.katgpu_last_error(self.engine.h).decode(errors="replace"))

    # ---- reducers ----
    def hist(self, low=1, high=10000, inc=1):
        base, ceil_, nb = hist_geometry(low, high)
        out = np.zeros(nb, np.uint64)
        self.engine._chk(self.engine.L.katgpu_hist(self.h, base, ceil_, inc, out.ctypes.data, nb))
        return out

    def gcp(self, cvg_scale=1.0, cvg_bins=1000):
        out = np.zeros((self.k, cvg_bins + 1), np.uint64)
        self.engine._chk(self.engine.L.katgpu_gcp(self.h, cvg_scale, cvg_bins, out.ctypes.data))
        return out

    # ---- multi-GPU exchange ----
    def partition_sizes(self, n_parts):
        out = np.zeros(n_parts, np.uint64)
        self.engine._chk(self.engine.L.katgpu_table_partition_sizes(self.h, n_parts, out.ctypes.data))
        return out

    def partition(self, n_parts, offsets, dev_keys_ptr, dev_counts_ptr):
        off = np.ascontiguousarray(offsets, np.uint64)
        self.engine._chk(self.engine.L.katgpu_table_partition(self.h, n_parts, off.ctypes.data, dev_keys_ptr, dev_counts_ptr))

    def merge_device(self, dev_keys_ptr, dev_counts_ptr, n):
        self.engine._chk(self.engine.L.katgpu_table_merge_device(self.h, dev_keys_ptr, dev_counts_ptr, n))

    # region-ordered exchange (dist.py)
    def slot_bytes(self):
        """HBM bytes per slot of this table: 8 (packed: remainder | count), 12 (KV12) or 20 (k > 32)."""
        return int(self.engine.L.katgpu_table_slot_bytes(self.h))

    def geometry(self):
        g = Geometry()
        self.engine._chk(self.engine.L.katgpu_table_geometry(self.h, C.byref(g)))
        return g

    def extract_sizes(self, n_parts, dev_region_counts_ptr):
        sizes = np.zeros(n_parts, np.uint64)
        self.engine._chk(self.engine.L.katgpu_table_extract_sizes(self.h, n_parts, dev_region_counts_ptr, sizes.ctypes.data))
        return sizes

    def extract(self, n_parts, dev_region_counts_ptr, dev_keys_ptr, dev_counts_ptr):
        """Returns the out-of-band records (counts above 32 bits, the all-ones k-mer) as (keys, counts) uint64 arrays."""
        bk, bc, nb = np.zeros(BIG_CAP, np.uint64), np.zeros(BIG_CAP, np.uint64), C.c_uint32()
        self.engine._chk(self.engine.L.katgpu_table_extract(self.h, n_parts, dev_region_counts_ptr, dev_keys_ptr, dev_counts_ptr,
                                                            bk.ctypes.data, bc.ctypes.data, BIG_CAP, C.byref(nb)))
        return bk[:nb.value].copy(), bc[:nb.value].copy()

    def packed_records(self):
        """True when the table can give 9-byte records (remainder + count): katgpu_table_extract_packed."""
        return bool(self.engine.L.katgpu_table_packed_records(self.h))

    def extract_packed(self, n_parts, dev_region_counts_ptr, dev_rem_lo_ptr, dev_rem_hi_ptr, dev_counts_ptr):
        bk, bc, nb = np.zeros(BIG_CAP, np.uint64), np.zeros(BIG_CAP, np.uint64), C.c_uint32()
        self.engine._chk(self.engine.L.katgpu_table_extract_packed(self.h, n_parts, dev_region_counts_ptr, dev_rem_lo_ptr, dev_rem_hi_ptr, dev_counts_ptr,
                                                                   bk.ctypes.data, bc.ctypes.data, BIG_CAP, C.byref(nb)))
        return bk[:nb.value].copy(), bc[:nb.value].copy()

    def merge_regions_packed(self, g_lo, g_hi, sources):
        """sources: (dev_rem_lo_ptr, dev_rem_hi_ptr, dev_counts_ptr, dev_region_counts_ptr, n_records, p1, p2) per sender."""
        arr = (MergeSourcePacked * len(sources))()
        for i, src in enumerate(sources):
            arr[i] = MergeSourcePacked(*src)
        self.engine._chk(self.engine.L.katgpu_table_merge_regions_packed(self.h, g_lo, g_hi, len(sources), C.cast(arr, C.c_void_p)))

    def clear(self):
        self.engine._chk(self.engine.L.katgpu_table_clear(self.h))

    def merge_device32(self, dev_keys_ptr, dev_counts_ptr, n):
        self.engine._chk(self.engine.L.katgpu_table_merge_device32(self.h, dev_keys_ptr, dev_counts_ptr, n))

    def merge_regions(self, g_lo, g_hi, sources):
        """sources: (dev_keys_ptr, dev_counts_ptr, dev_region_counts_ptr or None, n_records, p1, p2) per sender."""
        arr = (MergeSource * len(sources))()
        for i, (k, c, r, n, p1, p2) in enumerate(sources):
            arr[i] = MergeSource(k, c, r, n, p1, p2)
        self.engine._chk(self.engine.L.katgpu_table_merge_regions(self.h, g_lo, g_hi, len(sources), C.cast(arr, C.c_void_p)))

    def merge_host(self, keys, counts):
        k = np.ascontiguousarray(keys, np.uint64)
        c = np.ascontiguousarray(counts, np.uint64)
        assert k.size == c.size
        self.engine._chk(self.engine.L.katgpu_table_merge_host(self.h, k.ctypes.data, c.ctypes.data, k.size))


def comp(t1, t2, d1_scale=1.0, d2_scale=1.0, d1_bins=1001, d2_bins=1001):
    """Comp::compare + merge: (main matrix, 13 counters, 4 spectra)."""
    ss = min(d1_bins, d2_bins)
    mx = np.zeros((d1_bins, d2_bins), np.uint64)
    cc = np.zeros(13, np.uint64)
    sp = np.zeros((4, ss), np.uint64)
    e = t1.engine
    e._chk(e.L.katgpu_comp(t1.h, t2.h, int(t1.canonical), int(t2.canonical), d1_scale, d2_scale, d1_bins, d2_bins,
                           mx.ctypes.data, cc.ctypes.data, sp.ctypes.data))
    return mx, cc, sp


COMM_ID_BYTES = 256


class Comm:
    """The native multi-GPU communicator (include/katgpu.h "the exchange itself": kg_comm.hip).  One per process / GPU.
    rank 0 makes the id with Comm.unique_id() and hands it to the others by any side channel (a file, torch.distributed, MPI)."""

    @staticmethod
    def unique_id():
        L = load_library()
        buf = C.create_string_buffer(COMM_ID_BYTES)
        rc = L.katgpu_comm_unique_id(buf)
        if rc:
            raise KatGpuError(rc, "katgpu_comm_unique_id")
        return buf.raw

    def __init__(self, engine, rank, world, comm_id):
        self.engine = engine
        L = engine.L
        L.katgpu_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
        L.katgpu_comm_free.argtypes = [C.c_void_p]
        L.katgpu_comm_free.restype = None
        L.katgpu_comm_transport.argtypes = [C.c_void_p]
        L.katgpu_comm_transport.restype = C.c_char_p
        L.katgpu_comm_transport_note.argtypes = [C.c_void_p]
        L.katgpu_comm_transport_note.restype = C.c_char_p
        L.katgpu_comm_distinct_devices.argtypes = [C.c_void_p]
        L.katgpu_comm_barrier.argtypes = [C.c_void_p]
        L.katgpu_exchange_merge.argtypes = [C.c_void_p, C.c_void_p]
        L.katgpu_exchange_begin.argtypes = [C.c_void_p, C.c_void_p]
        L.katgpu_exchange_finish.argtypes = [C.c_void_p, C.c_void_p]
        L.katgpu_allreduce_u64.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.katgpu_comm_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 4 + [C.POINTER(C.c_uint64)] * 2
        L.katgpu_comm_wire.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        h = C.c_void_p()
        assert len(comm_id) == COMM_ID_BYTES
        engine._chk(L.katgpu_comm_init(engine.h, rank, world, comm_id, C.byref(h)))
        self.h, self.rank, self.world = h, rank, world

    @property
    def transport(self):
        return self.engine.L.katgpu_comm_transport(self.h).decode()

    @property
    def transport_note(self):
        return self.engine.L.katgpu_comm_transport_note(self.h).decode()

    @property
    def distinct_devices(self):
        """How many different devices the ranks run on (1: they all share one -- the /dev/shm transport's home ground)."""
        return int(self.engine.L.katgpu_comm_distinct_devices(self.h))

    def barrier(self):
        self.engine._chk(self.engine.L.katgpu_comm_barrier(self.h))

    def exchange_merge(self, table):
        """In place: afterwards `table` holds the k-mers this rank owns, counts summed over all ranks."""
        self.engine._chk(self.engine.L.katgpu_exchange_merge(self.h, table.h))
        return table

    def exchange_begin(self, table):
        """The exchange in two calls: the table's records go on the wire; count the next input, then exchange_finish(table)."""
        self.engine._chk(self.engine.L.katgpu_exchange_begin(self.h, table.h))
        return table

    def exchange_finish(self, table):
        self.engine._chk(self.engine.L.katgpu_exchange_finish(self.h, table.h))
        return table

    def allreduce_u64(self, arrays):
        """Sum uint64 numpy arrays over ranks; returns new arrays of the same shapes."""
        flat = np.concatenate([np.ascontiguousarray(a, np.uint64).reshape(-1) for a in arrays]) if arrays else np.zeros(0, np.uint64)
        self.engine._chk(self.engine.L.katgpu_allreduce_u64(self.h, flat.ctypes.data, flat.size))
        out, o = [], 0
        for a in arrays:
            n = int(np.prod(a.shape))
            out.append(flat[o:o + n].reshape(a.shape).copy())
            o += n
        return out

    def stats(self):
        d = [C.c_double() for _ in range(4)]
        u = [C.c_uint64() for _ in range(2)]
        self.engine._chk(self.engine.L.katgpu_comm_stats(self.h, *[C.byref(x) for x in d], *[C.byref(x) for x in u]))
        w = [C.c_uint64(), C.c_uint64()]
        pk = C.c_int()
        self.engine._chk(self.engine.L.katgpu_comm_wire(self.h, C.byref(w[0]), C.byref(w[1]), C.byref(pk)))
        return {"extract_ms": d[0].value, "exchange_ms": d[1].value, "merge_ms": d[2].value, "allreduce_ms": d[3].value,
                "bytes_sent": u[0].value, "merge_calls": u[1].value,
                "records_sent": w[0].value, "record_bytes_sent": w[1].value, "records_packed": bool(pk.value)}

    def free(self):
        if getattr(self, "h", None) and getattr(self.engine, "h", None):
            self.engine.L.katgpu_comm_free(self.h)
        self.h = None


def comp3(t1, t2, t3, d1_scale=1.0, d2_scale=1.0, d1_bins=1001, d2_bins=1001):
    """Three-input Comp::compare: (main, ends, middle, mixed, 13 counters, 4 spectra)."""
    ss = min(d1_bins, d2_bins)
    mxs = [np.zeros((d1_bins, d2_bins), np.uint64) for _ in range(4)]
    cc = np.zeros(13, np.uint64)
    sp = np.zeros((4, ss), np.uint64)
    e = t1.engine
    e._chk(e.L.katgpu_comp3(t1.h, t2.h, t3.h, int(t1.canonical), int(t2.canonical), int(t3.canonical), d1_scale, d2_scale,
                            d1_bins, d2_bins, mxs[0].ctypes.data, mxs[1].ctypes.data, mxs[2].ctypes.data, mxs[3].ctypes.data,
                            cc.ctypes.data, sp.ctypes.data))
    return mxs[0], mxs[1], mxs[2], mxs[3], cc, sp
