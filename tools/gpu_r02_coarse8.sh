#!/bin/bash
# 8 x 16 level on the chunked fp16x2 split kernel (half of every 32-pixel fragment row masked) vs the fp32 kernels
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
SGMSE_CHUNK_MIN_TILES=1 SGMSE_CHUNK_MIN_WIDTH=16 timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "split_k_of_the_coarse or tile_shape or forward_matches_reference_full_width or batch_independence" 2>&1 | tail -1
for cfg in "SGMSE_CHUNK_MIN_TILES=2" "SGMSE_CHUNK_MIN_TILES=1 SGMSE_CHUNK_MIN_WIDTH=16" "SGMSE_CHUNK_MIN_TILES=1 SGMSE_CHUNK_MIN_WIDTH=8"; do
  for b in 32 1; do
    env $cfg timeout 600 python bench.py --batch $b --steps $([ $b = 1 ] && echo 3 || echo 1) --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg batch $b:', round(d['ms_per_step']/1e3,4), 's per step', round(d['value'],3), 'utt/s', {k:(v['ms'],v['launches']) for k,v in d['kernel_classes_one_eval'].items() if k.startswith('conv3')})"
  done
done
