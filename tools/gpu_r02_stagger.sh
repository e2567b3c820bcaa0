#!/bin/bash
# Start-up stagger sweep of the fp16x2 3x3 split kernel (micro-benchmark, HIP events).  variant = 128 | mode << 22 | rows4 << 23 | units << 24
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
V="128"
for u in 1 2 3 4 6; do for m in 0 1; do V="$V,$((128 + (m<<22) + (u<<24)))"; done; done
V="$V,$((128 + (1<<23))),$((128 + (1<<23) + (1<<24))),$((128 + (1<<23) + (2<<24)))"
VARIANTS=$V SHAPES=${SHAPES:-0,1,2} FUSED=1 ROUNDS=3 OUT=split_stagger.json timeout 900 python tools/conv_microbench.py > gpurun_out/split_stagger.log 2>&1
echo "rc=$?"; cat gpurun_out/split_stagger.log
if [ -n "$BENCH_STAGGER" ]; then
  for s in $BENCH_STAGGER; do
    echo "== bench SGMSE_SPLIT_STAGGER=$s mode ${BENCH_MODE:-0}"
    SGMSE_SPLIT_STAGGER=$s SGMSE_SPLIT_STAGGER_MODE=${BENCH_MODE:-0} timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_stagger_$s.log 2>gpurun_out/bench_stagger_$s.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/bench_stagger_$s.log").read().strip().splitlines()[-1]); print("stagger $s", round(d["value"],3), "utt/s", round(d["roofline"]["frac"],4), {k:v["ms"] for k,v in d["kernel_classes_one_eval"].items()})
PY
  done
fi
