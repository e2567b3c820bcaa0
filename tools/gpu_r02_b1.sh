#!/bin/bash
# Batch-1 latency and batch-32 throughput after the coarse-level split-K / attention changes + the tests they touch
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "${TEST_K:-attention or forward or tile_shape or batch_independence or baseline_configuration or batch_of_four or samplers_match or long_utterance}" > gpurun_out/pytest_gpu_targeted2.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_targeted2.log
SGMSE_PROFILE_DUMP=1 timeout 600 python bench.py --batch 1 --N 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_b1.log 2> gpurun_out/prof_dump_b1_r02b.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_b1.log").read().strip().splitlines()[-1]); print("B=1 N=4 ms_per_step", round(d["ms_per_step"],1), "-> per evaluation", round(d["ms_per_step"]/8,2), "ms;", {k:(v["ms"],v["launches"]) for k,v in d["kernel_classes_one_eval"].items()})
PY
timeout 600 python bench.py --batch 1 --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=1 full N=30: s per utterance', round(d['ms_per_step']/1e3,3))"
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_b32.log 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_b32.log").read().strip().splitlines()[-1]); print("B=32", round(d["value"],3), "utt/s", round(d["roofline"]["frac"],4), {k:(v["ms"],v["launches"]) for k,v in d["kernel_classes_one_eval"].items()})
PY
