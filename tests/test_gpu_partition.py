"""The partitioned counter (kg_partition.hpp: two-level radix partition + regions applied in LDS) vs the oracle.
(tests/test_gpu_bench_geometry.py runs it at the bench's geometry against the direct kernel.)

Production thresholds send only >= 32 M-start inputs down this path; the test hooks force it for small inputs with
small regions / rounds so that multi-round accumulation, region spills, regrows and ragged tiles are all exercised at
sizes the oracle finishes in seconds.  (tests/test_gpu_scale_properties.py runs it at production settings.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


# grow_nomem: every table growth during a partition call "fails" beside the arena, so the spilled keys are parked on the
# host, the arena is given up and the call re-enters (the path a too-small size hint takes at production sizes).
# p2_fast=2: level 2 without its histogram pass (k_p2_fast) even on these tiny rounds, where runs overflow their capacity
# all the time (overflow list -> direct path); ovf_cap=3 makes that list overflow too (fall back to the exact kernel).
@pytest.mark.parametrize("region_slots,round_items,spill_mod,extra", [
    (512, 100000, 0, {}), (1024, 3000000, 7, {}), (8192, 400000, 0, {}),
    (512, 100000, 5, {"KATGPU_TEST_GROW_NOMEM": "1"}), (2048, 250000, 0, {"KATGPU_TEST_GROW_NOMEM": "1"}),
    (512, 100000, 0, {"KATGPU_P2_FAST": "2"}), (1024, 3000000, 7, {"KATGPU_P2_FAST": "2"}),
    (2048, 250000, 3, {"KATGPU_P2_FAST": "2", "KATGPU_TEST_P2_OVF_CAP": "3"}),
    (512, 150000, 0, {"KATGPU_P2_FAST": "2", "KATGPU_TEST_GROW_NOMEM": "1"}), (8192, 400000, 0, {"KATGPU_P2_FAST": "0"}),
    # l1_fast=2: level 1 without its counting pass (chunked scatter); L1_CPB caps the chunks of a bucket so that runs overflow
    # (list -> direct path) and, with a tiny list, the round is redone exactly
    (512, 100000, 0, {"KATGPU_L1_FAST": "2"}), (1024, 3000000, 7, {"KATGPU_L1_FAST": "2", "KATGPU_P2_FAST": "2"}),
    (8192, 400000, 0, {"KATGPU_L1_FAST": "2", "KATGPU_P2_FAST": "0"}),
    (2048, 250000, 0, {"KATGPU_L1_FAST": "2", "KATGPU_P2_FAST": "2", "KATGPU_TEST_L1_CPB": "3", "KATGPU_P1_WGS": "1"}),
    (512, 150000, 3, {"KATGPU_L1_FAST": "2", "KATGPU_TEST_L1_CPB": "2", "KATGPU_TEST_P2_OVF_CAP": "50", "KATGPU_P1_WGS": "1"}),
    (1024, 200000, 0, {"KATGPU_L1_FAST": "2", "KATGPU_P2_FAST": "2", "KATGPU_TEST_GROW_NOMEM": "1"}),
    # pass_buckets: level 2 + apply go through a round's buckets in passes of this many (production: one CU-full, and only at
    # sizes beyond these tests): the level-2 buffer holds one pass, every pass spills into its own part of the level-1 buffer, the
    # overflow list and the fall back to the exact level 2 carry over from pass to pass
    (512, 100000, 0, {"KATGPU_TEST_PASS_BUCKETS": "5"}), (1024, 3000000, 7, {"KATGPU_P2_FAST": "2", "KATGPU_TEST_PASS_BUCKETS": "3"}),
    (2048, 250000, 3, {"KATGPU_P2_FAST": "2", "KATGPU_TEST_P2_OVF_CAP": "3", "KATGPU_TEST_PASS_BUCKETS": "2"}),
    (512, 150000, 3, {"KATGPU_L1_FAST": "2", "KATGPU_TEST_L1_CPB": "2", "KATGPU_TEST_P2_OVF_CAP": "50", "KATGPU_P1_WGS": "1", "KATGPU_TEST_PASS_BUCKETS": "7"}),
    (1024, 200000, 0, {"KATGPU_L1_FAST": "2", "KATGPU_P2_FAST": "2", "KATGPU_TEST_GROW_NOMEM": "1", "KATGPU_TEST_PASS_BUCKETS": "1"}),
    # Slot layouts.  With these region sizes the k = 27 and k = 15 tables of the cases are packed (8-byte slots: remainder | count)
    # once they have 2^10 regions or more, the k = 31 / 32 ones and every small table KV12, and tables regrow from one into the other
    # in mid-call.  no_packed: every table KV12 (the apply, join and merge kernels of that layout).  ap_seg: the apply kernels walk
    # a run in segments of this many k-mers (production: 2^31 resp. half the in-slot count range) -- several segments per region.
    (512, 100000, 0, {"KATGPU_NO_PACKED": "1"}), (1024, 3000000, 7, {"KATGPU_NO_PACKED": "1", "KATGPU_P2_FAST": "2"}),
    (8192, 400000, 0, {"KATGPU_NO_PACKED": "1", "KATGPU_L1_FAST": "2"}),
    (512, 100000, 0, {"KATGPU_TEST_AP_SEG": "64"}), (2048, 250000, 0, {"KATGPU_TEST_AP_SEG": "1024", "KATGPU_P2_FAST": "2"}),
    (1024, 3000000, 0, {"KATGPU_TEST_AP_SEG": "256", "KATGPU_NO_PACKED": "1"}),
    (256, 3000000, 0, {}), (256, 3000000, 5, {"KATGPU_APPLY_PER_CU": "1"}),
    # l1_lean=0: level 1's ranking sweep and copy-out in their 64-bit form (the default is the 32-bit one of kg_l1_lean.hpp), both
    # editions of level 1
    (512, 100000, 0, {"KATGPU_L1_LEAN": "0"}), (1024, 3000000, 7, {"KATGPU_L1_LEAN": "0", "KATGPU_L1_FAST": "2"}),
    (8192, 400000, 0, {"KATGPU_L1_LEAN": "0", "KATGPU_L1_FAST": "2", "KATGPU_P2_FAST": "2"}),
    # lazy_min_slots=1: every packed table leaves the clearing of its slots to its first sweep -- the first round's apply starts every
    # region from zeros and visits all of them, pass by pass (production: tables of 2^26 slots and more); whatever else touches the table
    # first (the direct path, a regrow, a reducer) clears what has not been swept yet
    (512, 100000, 0, {"KATGPU_TEST_LAZY_MIN_SLOTS": "1"}), (1024, 3000000, 7, {"KATGPU_TEST_LAZY_MIN_SLOTS": "1", "KATGPU_P2_FAST": "2", "KATGPU_TEST_PASS_BUCKETS": "3"}),
    (2048, 250000, 3, {"KATGPU_TEST_LAZY_MIN_SLOTS": "1", "KATGPU_TEST_GROW_NOMEM": "1"}), (512, 100000, 0, {"KATGPU_TEST_LAZY_MIN_SLOTS": "1", "KATGPU_TEST_AP_SEG": "64"}),
    (256, 3000000, 5, {"KATGPU_TEST_LAZY_MIN_SLOTS": "1", "KATGPU_L1_FAST": "2", "KATGPU_P2_FAST": "2", "KATGPU_TEST_PASS_BUCKETS": "2"})])
def test_partitioned_counter_matches_oracle(region_slots, round_items, spill_mod, extra):
    env = dict(os.environ, KATGPU_PART_MIN_STARTS="0", KATGPU_TEST_REGION_SLOTS=str(region_slots),
               KATGPU_TEST_ROUND_ITEMS=str(round_items), KATGPU_TEST_SPILL_MOD=str(spill_mod))
    env.update(extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "partition_cases.py")], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "partition cases ok" in r.stdout
