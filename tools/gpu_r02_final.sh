#!/bin/bash
# Round-2 final validation on one MI355X: smoke, full GPU suite (with its printed error figures), kernel micro-benchmarks,
# PMC passes (MFMA-busy / clock, HBM FETCH / WRITE in separate passes), the bench line, the rocprofv3 kernel trace of the bench
# command and the other BASELINE workloads.  Everything lands under gpurun_out/ and is copied into profiles/ by hand.
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
STEPS=${STEPS:-smoke,pytest,micro,pmc,hbm,bench,rocprof,workloads,dumps}
has() { [[ ",$STEPS," == *",$1,"* ]]; }
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt; nproc >> gpurun_out/gpu.txt
if has smoke; then echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 | tee gpurun_out/smoke.log; fi
if has pytest; then echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log; fi
if has micro; then
  echo "== conv micro-benchmark: fp32 MFMA (0), bf16x3 (64), fp16x2 (128)"
  VARIANTS=0,64,128 SHAPES=0,1,2,3 FUSED=0,1 ROUNDS=3 OUT=conv_microbench.json timeout 600 python tools/conv_microbench.py 2>&1 | grep "^ks=" | tee gpurun_out/conv_microbench.log
fi
if has pmc; then
  echo "== PMC passes (dominant kernel, shape 0 and 1)"
  rm -rf gpurun_out/pmc
  VARIANTS=128 SHAPES=0,1 FUSED=1 ROUNDS=1 OUT=pmc_microbench.json timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc/p1 -o p -- python tools/conv_microbench.py > gpurun_out/pmc1.log 2>&1; echo "pmc1 rc=$?"
  VARIANTS=128 SHAPES=0,1 FUSED=1 ROUNDS=1 OUT=pmc_microbench.json timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc/p2 -o p -- python tools/conv_microbench.py > gpurun_out/pmc2.log 2>&1; echo "pmc2 rc=$?"
  python tools/summarize_pmc.py gpurun_out/pmc/p1 gpurun_out/pmc/p2 > gpurun_out/pmc_conv_split_h2.json; grep -E "effective_clock|mfma_pipe|grid=" gpurun_out/pmc_conv_split_h2.json | head -12
fi
if has hbm; then
  echo "== HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes; calibration stream + GroupNorm / FIR kernels in the same run)"
  rm -rf gpurun_out/hbm
  export VARIANTS=128 ROUNDS=1 SHAPES=0 FUSED=1 CALIB=1 OUT=hbm_microbench.json
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/hbm/fetch -o p -- python tools/conv_microbench.py > gpurun_out/hbm_fetch.log 2>&1; echo "fetch rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/hbm/write -o p -- python tools/conv_microbench.py > gpurun_out/hbm_write.log 2>&1; echo "write rc=$?"
  unset VARIANTS ROUNDS SHAPES FUSED CALIB OUT
  python tools/summarize_hbm.py gpurun_out/hbm/fetch gpurun_out/hbm/write gpurun_out/hbm_microbench.json > gpurun_out/hbm_traffic.json
  python - <<'PY'
import json
d = json.load(open("gpurun_out/hbm_traffic.json"))
for k, v in d.items():
    if isinstance(v, dict) and "hbm_bytes_per_launch" in v and any(n in k for n in ("conv3x3_split", "fir_", "gn_")):
        print(k[:80], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a in ("hbm_bytes_per_launch", "median_duration_us_under_pmc", "hbm_gb_per_s", "algorithmic_gb_per_s")})
PY
fi
if has bench; then echo "== bench (driver defaults)"; timeout 900 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/bench.log; tail -2 gpurun_out/bench.err; fi
if has rocprof; then
  echo "== rocprofv3 kernel trace of the bench command"
  rm -rf gpurun_out/prof
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_bench.log 2>gpurun_out/prof.err; echo "rocprof rc=$?"
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-220
  find gpurun_out/prof -name "*kernel_trace.csv" -size +30M -delete
fi
if has workloads; then
  echo "== other BASELINE workloads"
  : > gpurun_out/other_workloads.txt
  for args in "--workload ode16k --steps 2 --warmup 1" "--workload pc48k --steps 1 --warmup 1" "--batch 1 --steps 3 --warmup 1" "--batch 8 --steps 2 --warmup 1"; do
    timeout 900 python bench.py $args --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/other_workloads.txt
    tail -1 gpurun_out/other_workloads.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$args ->', round(d['value'],3), d['unit'], 'ms_per_step', round(d['ms_per_step'],1), 'rtf', round(d['rtf'],4), 'roofline frac', round(d.get('roofline',{}).get('frac',0),4))"
  done
fi
if has dumps; then
  echo "== per-launch timings of one evaluation (batch 32 and batch 1)"
  for b in 32 1; do
    SGMSE_PROFILE_DUMP=1 timeout 600 python bench.py --batch $b --N 2 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_dump_b${b}_r02_final.txt
    grep -c sgmse-prof gpurun_out/prof_dump_b${b}_r02_final.txt
  done
fi
