#!/bin/bash
# One GPU-box visit: smoke, GPU parity tests, conv micro-benchmark (+PMC), bench, rocprofv3 kernel trace.
# Everything is logged under gpurun_out/ (merged back by gpurun).  Steps are selected by STEPS (default: all).
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
STEPS=${STEPS:-smoke,pytest,micro,pmc,bench,rocprof}
has() { [[ ",$STEPS," == *",$1,"* ]]; }
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt; nproc >> gpurun_out/gpu.txt
if has smoke; then echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log; fi
if has pytest; then echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu.log; fi
if has micro; then echo "== conv microbench"; timeout 900 python tools/conv_microbench.py > gpurun_out/conv_microbench.log 2>&1; echo "micro rc=$?"; cat gpurun_out/conv_microbench.log | tail -80; fi
if has pmc; then
  echo "== PMC (conv microbench, separate counter passes)"
  rm -rf gpurun_out/pmc
  VARIANTS=${PMC_VARIANTS:-0,1} ROUNDS=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc/p1 -o p -- python tools/conv_microbench.py > gpurun_out/pmc1.log 2>&1; echo "pmc1 rc=$?"
  VARIANTS=${PMC_VARIANTS:-0,1} ROUNDS=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc/p2 -o p -- python tools/conv_microbench.py > gpurun_out/pmc2.log 2>&1; echo "pmc2 rc=$?"
  tail -3 gpurun_out/pmc1.log; ls -la gpurun_out/pmc/p1 gpurun_out/pmc/p2 2>/dev/null | head
fi
if has bench; then echo "== bench"; timeout 900 python bench.py --steps ${BENCH_STEPS:-1} --warmup 1 > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.log; tail -3 gpurun_out/bench.err; fi
for v in ${BENCH_VARIANTS:-}; do echo "== bench SGMSE_CONV_VARIANT=$v"; SGMSE_CONV_VARIANT=$v timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_v$v.log 2>gpurun_out/bench_v$v.err; cat gpurun_out/bench_v$v.log | cut -c1-200; python -c "
import json,sys
d=json.loads(open('gpurun_out/bench_v$v.log').read().strip().splitlines()[-1]); print('variant $v', d['value'], d['roofline']['achieved'], {k:v['ms'] for k,v in d['kernel_classes_one_eval'].items()})"; done
if has sweep; then
  echo "== tile_min_blocks sweep (short sampler, N=6)"
  for b in ${SWEEP_BATCHES:-1 4 32}; do for m in ${SWEEP_MINBLK:-256 512 1024 2048}; do
    SGMSE_TILE_MIN_BLOCKS=$m timeout 300 python bench.py --batch $b --N 6 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/sweep_b${b}_m$m.log 2>/dev/null
    python -c "
import json
d=json.loads(open('gpurun_out/sweep_b${b}_m$m.log').read().strip().splitlines()[-1]); print('batch $b min_blocks $m ms_per_step', round(d['ms_per_step'],1), {k:round(v['ms'],2) for k,v in d['kernel_classes_one_eval'].items()})"
  done; done 2>&1 | tee gpurun_out/sweep.txt
fi
if has hbm; then
  echo "== HBM traffic of the dominant kernel (FETCH_SIZE / WRITE_SIZE passes; rocprofv3 --pmc segfaults on the full bench command)"
  rm -rf gpurun_out/hbm
  export VARIANTS=${HBM_VARIANTS:-128} ROUNDS=1 SHAPES=0 FUSED=1 CALIB=1 OUT=hbm_microbench.json
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/hbm/fetch -o p -- python tools/conv_microbench.py > gpurun_out/hbm_fetch.log 2>&1; echo "fetch rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/hbm/write -o p -- python tools/conv_microbench.py > gpurun_out/hbm_write.log 2>&1; echo "write rc=$?"
  unset VARIANTS ROUNDS SHAPES FUSED CALIB OUT
  python tools/summarize_hbm.py gpurun_out/hbm/fetch gpurun_out/hbm/write gpurun_out/hbm_microbench.json > gpurun_out/hbm_traffic.json; head -c 1800 gpurun_out/hbm_traffic.json
fi
if has rocprof; then
  echo "== rocprofv3 kernel trace of the bench command"
  rm -rf gpurun_out/prof
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o trace -- python bench.py --steps 1 --warmup 0 --batch ${PROF_BATCH:-32} --no-cpu-baseline > gpurun_out/prof_bench.log 2>gpurun_out/prof.err; echo "rocprof rc=$?"
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
  find gpurun_out/prof -name "*kernel_trace.csv" -size +30M -delete
fi
