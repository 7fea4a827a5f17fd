"""kg_l1_lean.hpp (level 1's per-window arithmetic on 32-bit halves) against a naive restatement and the placement hash of
kg_device.hpp, on the host: tests/native/l1_lean_check.cc is built with hipcc (-x hip, so that it includes the real headers) and run
on the CPU -- 1.3 M windows over k = 17 .. 31, both strand modes, seven level-1 digit counts, flag patterns from none to dense."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lean_arithmetic_matches_naive(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "l1_lean_check")
    r = subprocess.run([hipcc, "-x", "hip", "--offload-arch=gfx950", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "native", "l1_lean_check.cc"), "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "l1 lean ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
