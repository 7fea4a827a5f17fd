"""The device-side record scan of raw FASTQ / FASTA bytes (kg_scan.hip / kg_scan.hpp) behind katgpu_count_files, against the host
state machine's stream counted by the oracle.  The hooks make batches tiny so that the files of tests/scan_cases.py cross hundreds of
batch cuts (record-aligned for FASTQ, line-aligned -- in the middle of a record's sequence -- for FASTA), and force the hand-over
to the host parser at a given batch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("extra", [
    {"KATGPU_TEST_SCAN_BATCH": "16384", "KATGPU_TEST_SCAN_SEGMENT": "4096", "KATGPU_TEST_SCAN_OVERLAP": "2048"},
    {"KATGPU_TEST_SCAN_BATCH": "5000", "KATGPU_TEST_SCAN_SEGMENT": "1000", "KATGPU_TEST_SCAN_OVERLAP": "1500", "KATGPU_SCAN_THREADS": "3"},
    {"KATGPU_TEST_SCAN_BATCH": "1048576", "KATGPU_TEST_SCAN_SEGMENT": "65536", "KATGPU_TEST_SCAN_OVERLAP": "4096"},
    {"KATGPU_TEST_SCAN_BATCH": "20000", "KATGPU_TEST_SCAN_SEGMENT": "20000", "KATGPU_TEST_SCAN_OVERLAP": "4096", "KATGPU_TEST_SCAN_FAIL_AT": "3"},   # host parser from batch 3 on
    {"KATGPU_TEST_SCAN_BATCH": "16384", "KATGPU_TEST_SCAN_OVERLAP": "2048", "KATGPU_PART_MIN_STARTS": "0", "KATGPU_TEST_REGION_SLOTS": "512",
     "KATGPU_TEST_ROUND_ITEMS": "100000"},                                                    # the scan's chunks through partition rounds
    # FASTQ stripped to its sequence lines by the readers (the production path of large FASTQ files); FASTA keeps the device scan
    {"KATGPU_TEST_STRIP_SEGMENT": "4096", "KATGPU_TEST_STRIP_OVERLAP": "2048", "KATGPU_TEST_SCAN_BATCH": "16384", "KATGPU_TEST_SCAN_OVERLAP": "2048"},
    {"KATGPU_TEST_STRIP_SEGMENT": "1000", "KATGPU_TEST_STRIP_OVERLAP": "1500", "KATGPU_SCAN_THREADS": "3", "KATGPU_TEST_SCAN_BATCH": "5000", "KATGPU_TEST_SCAN_OVERLAP": "1500"},
    {"KATGPU_TEST_STRIP_SEGMENT": "65536", "KATGPU_TEST_STRIP_OVERLAP": "4096", "KATGPU_TEST_SCAN_BATCH": "1048576", "KATGPU_TEST_SCAN_OVERLAP": "4096", "KATGPU_SCAN_MMAP": "1"},
    {"KATGPU_TEST_STRIP_SEGMENT": "8192", "KATGPU_TEST_STRIP_OVERLAP": "4096", "KATGPU_TEST_STRIP_FAIL_AT": "5", "KATGPU_TEST_SCAN_BATCH": "20000", "KATGPU_TEST_SCAN_OVERLAP": "4096"},   # host parser from segment 5 on
    {"KATGPU_TEST_STRIP_SEGMENT": "4096", "KATGPU_TEST_STRIP_OVERLAP": "2048", "KATGPU_TEST_SCAN_BATCH": "16384", "KATGPU_TEST_SCAN_OVERLAP": "2048", "KATGPU_PART_MIN_STARTS": "0",
     "KATGPU_TEST_REGION_SLOTS": "512", "KATGPU_TEST_ROUND_ITEMS": "100000", "KATGPU_TEST_SCAN_ACC": "40000"}])       # many small accumulation buffers through partition rounds
def test_device_scan_matches_host_parser(extra):
    env = dict(os.environ)
    env.update(extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "scan_cases.py")], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "scan cases ok" in r.stdout
