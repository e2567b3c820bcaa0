#!/bin/bash
# Phase experiments on the fp16x2 3x3 split kernel: epilogue variants and per-workgroup phase time stamps.
# variant = 128 | abl << 12 (abl: 1 no stores, 2 no residual read, 64 trace, 128 residual ring depth 1, 256 nt policy, 512 one WG per CU)
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
V="128"; for a in 1 2 3 128 256 512 64; do V="$V,$((128 + (a<<12)))"; done
SGMSE_TRACE_OUT=gpurun_out/trace_split.bin VARIANTS=$V SHAPES=${SHAPES:-0,1} FUSED=1 ROUNDS=3 OUT=split_phases.json timeout 900 python tools/conv_microbench.py > gpurun_out/split_phases.log 2>&1
echo "rc=$?"; cat gpurun_out/split_phases.log
python tools/analyze_trace.py gpurun_out/trace_split.bin | tee gpurun_out/trace_split.txt | head -60
