"""The host / file entry points (katgpu_count, katgpu_count_files, katgpu_count_bases_host -- what replaces
InputHandler::count, lib/src/input_handler.cc:180-202) run the PARTITIONED counter: pinned staging -> device rings -> partition
rounds, bit-identical to the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


# (a ring must hold more than 1 M window starts, or the partitioned counter declines it as not worth a round)
@pytest.mark.parametrize("ring_mb,extra", [(2, {}), (3, {"KATGPU_TEST_REGION_SLOTS": "1024"}), (2, {"KATGPU_APPLY_V": "1"})])
def test_rings_through_the_partitioned_counter(ring_mb, extra):
    env = dict(os.environ, KATGPU_RING_MB=str(ring_mb), KATGPU_PART_MIN_STARTS="0")
    env.update(extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "feeder_cases.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "feeder cases ok" in r.stdout


def test_count_files_production_settings(engine, ko, tmp_path):
    """A file large enough for the production threshold (>= 32 M window starts per ring): engine.count(paths) must have run
    partition rounds, and the table must equal the oracle's."""
    from kat_amd import synth
    g = synth.genome(2_000_000, seed=21)
    reads = synth.reads(g, 0, 300_000, seed=22).reshape(-1, 151)[:, :150]        # 45 M bases
    p = tmp_path / "big.fq"
    rec = np.empty((reads.shape[0], 318), np.uint8)
    rec[:, :14] = np.frombuffer(b"@r000000000/1\n", np.uint8)
    rec[:, 14:164] = reads
    rec[:, 164:167] = np.frombuffer(b"\n+\n", np.uint8)
    rec[:, 167:317] = ord("I")
    rec[:, 317] = ord("\n")
    p.write_bytes(rec.tobytes())
    engine.profile_reset()
    gt = engine.count([str(p)], 27, True)
    prof = engine.profile()
    assert prof["part_apply"]["launches"] > 0 and prof["part_l1_scatter"]["launches"] > 0, prof
    ot = ko.Table(27, True).count_files([str(p)])
    gk, gc = gt.dump_sorted()
    ok_, oc = ot.dump_sorted()
    assert np.array_equal(gk, ok_) and np.array_equal(gc, oc)
