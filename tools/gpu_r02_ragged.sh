#!/bin/bash
# Ragged batches on the GPU: the parity tests the change touches (kernel families now follow the U-Net level: utterances of
# other lengths than 512 frames take other kernels than before), the new ragged test, a short bench at batch 32 and batch 1
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "ragged or forward_matches or long_utterance or full_width_48k or full_size_forward or tile_shape or split_k_of or batch_independence or samplers_match or adversarial or enhancement_script or do_not_depend or sampler_graph or front_end" > gpurun_out/pytest_gpu_ragged.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu_ragged.log
for b in 32 1; do
  timeout 600 python bench.py --batch $b --steps $([ $b = 1 ] && echo 3 || echo 1) --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b:', round(d['ms_per_step']/1e3,4), 's per step', round(d['value'],3), 'utt/s', 'frac', round(d['roofline']['frac'],4))"
done
