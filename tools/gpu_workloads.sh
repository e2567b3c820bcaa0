#!/bin/bash
# Other BASELINE.json configurations on one GPU: configs[2] (fixed-step PF-ODE), configs[3] (48 kHz), and the single-utterance
# latency case (configs[0] shape on the GPU).
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
for spec in "ode16k --steps 1 --warmup 1" "pc16k --batch 1 --steps 2 --warmup 1" "pc16k --batch 8 --steps 1 --warmup 1" "pc48k --steps 1 --warmup 0"; do
  name=$(echo $spec | tr ' -' '__')
  echo "== bench --workload $spec"
  timeout 900 python bench.py --workload $spec --no-cpu-baseline > gpurun_out/bench_$name.log 2>gpurun_out/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/bench_$name.log').read().strip().splitlines()[-1])
    print({k:d[k] for k in ('value','ms_per_step','rtf','path_tflops')}, d['config']['workload'], d['config'].get('arena_gb'), d['roofline']['achieved'])
except Exception as e:
    print('FAILED', e); print(open('gpurun_out/bench_$name.err').read()[-1500:])
PY
done
