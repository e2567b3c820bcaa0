#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
for cfg in "SGMSE_COARSE_SPLIT=0" "SGMSE_COARSE_SPLITK_DIV=100000" "SGMSE_COARSE_SPLITK_DIV=8"; do
  for b in 32 1; do
    env $cfg timeout 600 python bench.py --batch $b --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg batch $b:', round(d['ms_per_step']/1e3,3), 's per step', round(d['value'],3), 'utt/s', {k:(v['ms'],v['launches']) for k,v in d['kernel_classes_one_eval'].items() if k.startswith('conv3')})"
  done
done
