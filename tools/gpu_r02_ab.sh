#!/bin/bash
# A/B on one box: main's build (ab_main/, before the ragged-batch changes) vs this tree, batch 32, alternating
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
for rep in 1 2; do
  for d in ab_main .; do
    (cd $d && timeout 200 python bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d rep $rep:', round(d['ms_per_step']/1e3,4), 's per step', round(d['value'],3), 'utt/s')")
  done
done
