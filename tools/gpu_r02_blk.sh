#!/bin/bash
# 2x4 register blocking of the split kernel vs 1x8 (micro-benchmark + bench), and a per-launch profile of one B = 1 evaluation
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for blk in 0 1; do
  echo "== SGMSE_SPLIT_BLK=$blk"
  SGMSE_SPLIT_BLK=$blk VARIANTS=128 SHAPES=0,1,2 FUSED=1 ROUNDS=3 OUT=split_blk$blk.json timeout 300 python tools/conv_microbench.py 2>&1 | grep "^ks="
done
SGMSE_SPLIT_BLK=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_blk1.log 2>gpurun_out/bench_blk1.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_blk1.log").read().strip().splitlines()[-1]); print("blk1", round(d["value"],3), "utt/s", round(d["roofline"]["frac"],4), {k:v["ms"] for k,v in d["kernel_classes_one_eval"].items()})
PY
echo "== B=1 per-launch profile"
SGMSE_PROFILE_DUMP=1 timeout 600 python bench.py --batch 1 --N 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_b1.log 2> gpurun_out/prof_dump_b1_r02.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_b1.log").read().strip().splitlines()[-1]); print("B=1 N=4 ms_per_step", round(d["ms_per_step"],1), "-> per evaluation", round(d["ms_per_step"]/8,2), "ms;", {k:(v["ms"],v["launches"]) for k,v in d["kernel_classes_one_eval"].items()})
PY
timeout 600 python bench.py --batch 1 --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=1 full N=30: s per utterance', round(d['ms_per_step']/1e3,3))"
