#!/bin/bash
# Per-phase / per-tap time stamps of the fp16x2 3x3 split kernel (ABL 64), also without epilogue memory traffic (67) and without staging (72)
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for a in 64 67 72; do
  SGMSE_TRACE_OUT=gpurun_out/trace_split_$a.bin VARIANTS=$((128 + (a<<12))) SHAPES=${SHAPES:-0} FUSED=1 ROUNDS=2 OUT=split_trace_$a.json timeout 300 python tools/conv_microbench.py 2>&1 | grep "^ks="
  python tools/analyze_trace.py gpurun_out/trace_split_$a.bin > gpurun_out/trace_split_$a.txt; head -24 gpurun_out/trace_split_$a.txt
done
