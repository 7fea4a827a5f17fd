"""kat_amd -- MI355X-native engine for the `kat hist` / `kat gcp` / `kat comp` hot path.

csrc/      HIP kernels + C ABI (libkatgpu.so, declared in include/katgpu.h) + the C++ host mirror of KAT's
           InputHandler / Histogram / Gcp / Comp (csrc/host, built into bin/katgpu)
binding.py ctypes binding of the C ABI
synth.py   seeded synthetic genome / read generator (host edition of the device generator)
dist.py    one-process-per-GPU sharding; the owner-partitioned merge through the native communicator (binding.Comm -> kg_comm.hip: RCCL,
           /dev/shm as the fall-back) or, for the CPU stand-in tests and as bench.py's last resort, over torch.distributed
"""
from .binding import Engine, Table, DeviceBuffer, Comm, KatGpuError, comp, comp3, hist_geometry, load_library, parse_file, parse_files, jf_read_records, jf_write_records, jf_read_records_wide, jf_write_records_wide, LIB_PATH, EXPORTS  # noqa: F401
