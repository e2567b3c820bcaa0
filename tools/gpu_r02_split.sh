#!/bin/bash
# Round-2 GPU visit for the dominant kernel: targeted parity tests, ablation micro-benchmark of the fp16x2 3x3 split kernel
# (+ PMC passes: effective clock, MFMA-busy), bench line.  Logs under gpurun_out/.
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
STEPS=${STEPS:-tests,abl,pmc,bench}
has() { [[ ",$STEPS," == *",$1,"* ]]; }
# variant = 128 (fp16x2 split kernel) | ablation << 12: 0 full, 3 no epilogue stores/residual, 4 no exp/rcp, 8 nothing staged
# after stage 0, 24 + B fragments read once, 56 + A fragments loaded once (MFMA-only loop), 59 + no epilogue memory traffic
ABL_VARIANTS="128,$((128 + (3<<12))),$((128 + (4<<12))),$((128 + (8<<12))),$((128 + (24<<12))),$((128 + (56<<12))),$((128 + (59<<12)))"
if has tests; then
  echo "== targeted pytest -m gpu"
  timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -s -k "${TEST_K:-conv or groupnorm or forward or split or baseline_configuration or batch_of_four or tile_shape or sampler}" > gpurun_out/pytest_gpu_targeted.log 2>&1
  echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu_targeted.log; grep -E "rel_l2 vs the reference|error vs fp64" gpurun_out/pytest_gpu_targeted.log | tail -30
fi
if has abl; then
  echo "== ablation micro-benchmark (HIP events, unprofiled)"
  VARIANTS=$ABL_VARIANTS SHAPES=${ABL_SHAPES:-0,1,2} FUSED=1 ROUNDS=3 OUT=split_ablation.json timeout 900 python tools/conv_microbench.py > gpurun_out/split_ablation.log 2>&1
  echo "abl rc=$?"; cat gpurun_out/split_ablation.log
fi
if has pmc; then
  echo "== PMC passes over the ablation variants (shape 0)"
  rm -rf gpurun_out/pmc
  VARIANTS=$ABL_VARIANTS SHAPES=0 FUSED=1 ROUNDS=1 OUT=pmc_microbench.json timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc/p1 -o p -- python tools/conv_microbench.py > gpurun_out/pmc1.log 2>&1; echo "pmc1 rc=$?"
  VARIANTS=$ABL_VARIANTS SHAPES=0 FUSED=1 ROUNDS=1 OUT=pmc_microbench.json timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc/p2 -o p -- python tools/conv_microbench.py > gpurun_out/pmc2.log 2>&1; echo "pmc2 rc=$?"
  python tools/summarize_pmc.py gpurun_out/pmc/p1 gpurun_out/pmc/p2 > gpurun_out/pmc_split_ablation.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_split_ablation.json"))
for k, v in d.items():
    print(k[:110], {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk != "counters_mean"})
PY
  find gpurun_out/pmc -name "*.csv" -size +20M -delete
fi
if has bench; then
  echo "== bench"
  timeout 900 python bench.py --steps ${BENCH_STEPS:-1} --warmup 1 ${BENCH_ARGS:-} > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.log; tail -3 gpurun_out/bench.err
fi
