#!/bin/bash
# round 3, visit A: GPU parity tests + headline bench after the data-driven fp16x2 input scale
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider > gpurun_out/r03a_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03a_pytest_gpu.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r03a_bench.json
