#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_ingest_at_size.py "tests/test_gpu_cli.py::test_gpus_switch_writes_the_single_gpu_files" -q -x --timeout=600 --durations=8 -p no:cacheprovider 2>&1 | tail -50) > gpurun_out/r05_striptests.log 2>&1
tail -22 gpurun_out/r05_striptests.log | cut -c1-250
(timeout 900 python tools/e2e_ab.py --reads 150000000 - KATGPU_FASTQ_STRIP=0 KATGPU_SCAN_THREADS=32 KATGPU_SCAN_THREADS=24 - 2>&1 | tail -60) > gpurun_out/r05_e2e_ab.txt; cat gpurun_out/r05_e2e_ab.txt | cut -c1-330
