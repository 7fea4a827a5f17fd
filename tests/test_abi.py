"""The C-ABI library loads here (no GPU) and exports every symbol include/katgpu.h declares; without a device the
product refuses to run instead of falling back to anything."""
import ctypes
import os
import re
import subprocess

import pytest

import kat_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "katgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(katgpu_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported():
    names = declared_symbols()
    assert len(names) >= 30
    lib = ctypes.CDLL(kat_amd.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(kat_amd.EXPORTS) == names          # the ctypes binding covers exactly the header


def test_no_reference_or_oracle_symbols_in_product():
    out = subprocess.run(["nm", "-D", "--defined-only", kat_amd.LIB_PATH], capture_output=True, text=True).stdout
    assert "ko_" not in out and "jellyfish" not in out.lower()
    ldd = subprocess.run(["ldd", kat_amd.LIB_PATH], capture_output=True, text=True).stdout
    assert "koracle" not in ldd and "libamdhip64" in ldd


def test_version_and_host_only_entry_points():
    L = kat_amd.load_library()
    assert b"gfx950" in L.katgpu_version()


@pytest.mark.skipif(os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK), reason="a GPU is visible")
def test_no_cpu_fallback():
    with pytest.raises(kat_amd.KatGpuError) as ei:
        kat_amd.Engine(0)
    assert ei.value.code == 8
    exe = os.path.join(ROOT, "kat_amd", "bin", "katgpu")
    if os.path.exists(exe):
        r = subprocess.run([exe, "hist", "-m", "27", os.path.join(ROOT, "tests", "golden", "refdata", "sect_test.fa")],
                           capture_output=True, text=True, cwd=os.environ.get("TMPDIR", "/tmp"))
        assert r.returncode == 5 and "no gfx950 device" in r.stderr


@pytest.mark.skipif(os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK), reason="a GPU is visible")
def test_gpus_launcher_fails_fast_without_a_device(tmp_path):
    """`katgpu <mode> --gpus N` forks its ranks before any HIP call: without a device every rank fails at katgpu_init and the launcher
    returns that exit code at once (no rank waits for an id nobody will send); the option's range is checked before anything runs."""
    exe = os.path.join(ROOT, "kat_amd", "bin", "katgpu")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    fq = os.path.join(ROOT, "tests", "golden", "refdata", "ecoli_r1.1K.fastq")
    r = subprocess.run([exe, "hist", "--gpus", "2", "-m", "27", "-o", "x.hist", fq], capture_output=True, text=True, cwd=tmp_path, timeout=60)
    assert r.returncode == 5 and "no gfx950 device" in r.stderr
    assert not (tmp_path / "x.hist").exists()
    r = subprocess.run([exe, "hist", "--gpus", "300", "-m", "27", fq], capture_output=True, text=True, cwd=tmp_path, timeout=60)
    assert r.returncode == 1 and "--gpus takes 1 .. 256" in r.stderr
    r = subprocess.run([exe, "comp", "--gpus=0", "-m", "27", fq, fq], capture_output=True, text=True, cwd=tmp_path, timeout=60)
    assert r.returncode == 1 and "--gpus" in r.stderr
