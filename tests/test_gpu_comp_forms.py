"""`kat comp` (katgpu_comp) through each of its kernel forms against the oracle.  The form follows from the tables (same region grid?
packed slots? both canonical?); the hooks below switch the preferred forms off one by one so that every one of them carries the
whole case list of tests/comp_cases.py, and shrink the regions so that tables of test size have enough of them to be packed."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("extra", [
    {},                                                                     # fused join where it applies, else two-pass join / probes
    {"KATGPU_NO_FUSED": "1"},                                               # packed two-pass join + the seen-bit pass 2
    {"KATGPU_NO_FUSED": "1", "KATGPU_NO_SEEN": "1", "KATGPU_FORCE_JOIN": "1", "KATGPU_NO_FOLD": "1", "KATGPU_JOIN_BLOCK": "1024"},   # join form for pass 2 as well
    {"KATGPU_NO_PACKED": "1"},                                              # KV12 slots: the 12-byte join
    {"KATGPU_NO_JOIN": "1"},                                                # HBM probes only
    {"KATGPU_TEST_REGION_SLOTS": "9000"},                                   # regions of the size the large tables have: the fused join's five-pair shape
    {"KATGPU_COMP_PLAIN_INC": "0"}])                                        # LDS increments aggregated per wave (ballots inside the queue drains)
def test_comp_forms_match_oracle(extra):
    env = dict(os.environ, KATGPU_TEST_REGION_SLOTS="512")
    env.update(extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "comp_cases.py")], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "comp cases ok" in r.stdout
