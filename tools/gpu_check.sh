#!/bin/bash
# One GPU-box visit: smoke, GPU parity tests, bench (+ A/B of the conv register target), rocprofv3 kernel trace.
# Everything is logged under gpurun_out/ (merged back by gpurun).
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py --steps ${BENCH_STEPS:-1} --warmup 1 > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.log; tail -5 gpurun_out/bench.err
if [ -n "$AB_MINW" ]; then
  echo "== bench MINW=1"; SGMSE_CONV_MINW=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_minw1.log 2>gpurun_out/bench_minw1.err; cat gpurun_out/bench_minw1.log
fi
echo "== rocprofv3"
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o trace -- python bench.py --steps 1 --warmup 0 --batch ${PROF_BATCH:-8} --no-cpu-baseline > gpurun_out/prof_bench.log 2>gpurun_out/prof.err; echo "rocprof rc=$?"
find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
# keep the merged-back payload small: drop the raw per-dispatch trace, keep the stats
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
