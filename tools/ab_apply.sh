#!/bin/bash
# A/B of the apply kernel editions on one box: the count stage's per-kernel HIP-event times.
# usage (GPU box): bash tools/ab_apply.sh "<variants: 'V UNR' ...>" [reads] [genome]
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
export KATGPU_TESTING=1     # the A/B switches are test hooks
R=${2:-300000000}; G=${3:-1000000000}
IFS=';' read -ra VARS <<< "${1:-1 83;2 83}"
for v in "${VARS[@]}"; do
  set -- $v
  echo "== KATGPU_APPLY_V=$1 KATGPU_APPLY_UNR=$2"
  KATGPU_APPLY_V=$1 KATGPU_APPLY_UNR=$2 timeout 240 python bench.py --reads $R --genome $G --steps 2 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step', d['ms_per_step'], 'ok', d['result_accounts_for_every_kmer'], d['kernel_ms_per_step'])
    elif l: print(l[:300])
"
done
