#!/bin/bash
# Entry convolution on the MFMA kernel (+ batched loads in the direct kernel): parity tests it touches, batch 32 / batch 1
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "forward or tile_shape or batch_independence or baseline_configuration or samplers_match or adversarial or weight_reload" 2>&1 | tail -2
for cfg in ${CFGS:-SGMSE_ENTRY_MFMA=0 SGMSE_ENTRY_MFMA=1}; do
  for b in 32 1; do
    env $cfg SGMSE_PROFILE_DUMP=1 timeout 600 python bench.py --batch $b --steps $([ $b = 1 ] && echo 3 || echo 1) --warmup 1 --no-cpu-baseline 2>gpurun_out/entry_dump_${cfg#*=}_b$b.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg batch $b:', round(d['ms_per_step']/1e3,4), 's per step', round(d['value'],3), 'utt/s', {k:(v['ms'],v['launches']) for k,v in d['kernel_classes_one_eval'].items()})"
    grep -m3 "@${b}x256x512" gpurun_out/entry_dump_${cfg#*=}_b$b.txt | cut -c1-120
  done
done
