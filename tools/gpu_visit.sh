#!/bin/bash
# One parameterised GPU visit (replaces the per-visit scripts of rounds 2-3).  Runs on the gpurun box from the repo root:
#   gpurun --timeout 900 -- 'STEPS=pytest,bench TAG=r04 bash tools/gpu_visit.sh'
# STEPS (comma list, run in this order):
#   smoke      __graft_entry__.smoke()
#   pytest     pytest -m gpu (PYTEST_K = -k expression, PYTEST_X=1 stops at the first failure)
#   micro      tools/conv_microbench.py (VARIANTS / SHAPES / FUSED / ROUNDS as that script reads them)
#   trace      phase time stamps of a traced kernel instantiation (TRACE_VARIANTS, TRACE_SHAPE) through tools/analyze_trace.py
#   power      tools/power_probe.py (CASES, ITERS)
#   ab         bench.py A/B: AB = ';'-separated "name|bench args|ENV=1 ENV2=0" entries
#   bench      bench.py with the driver's defaults -> ${TAG}_bench_b32.json
#   rocprof    rocprofv3 --kernel-trace --stats of the bench command -> ${TAG}_bench_b32_kernel_stats.csv
#   pmc        PMC passes of the dominant kernel's micro-benchmark (tools/conv_microbench.py) -> ${TAG}_pmc_*.json, ${TAG}_hbm_traffic_*.json
#   pmcbench   PMC passes over the bench command itself at batch 32 (counters of the run's own launches of the dominant kernel)
#              -> ${TAG}_pmc_bench_b32.json (bench.py reads it: roofline.traffic, roofline.mfma_busy)
#   dumps      per-launch timings of one evaluation at batch 32 and batch 1
#   ragged     tools/ragged_bench.py
#   dirjob     tools/dir_job_bench.py
#   lengths    bench.py --seconds sweep (utterance lengths off the 512-frame grid)
#   libab      same-box A/B of TWO builds of the library: LIB_A (default: the product library) and LIB_B (a second build placed in the
#              tree, e.g. from a scratch copy of csrc/), in the order A B B A: the kernel micro-benchmark (VARIANTS / SHAPES) and
#              bench.py at batch 32 and batch 1
# Everything lands in gpurun_out/ with the TAG prefix; copy what is to be judged into profiles/.
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
TAG=${TAG:-r04}
STEPS=${STEPS:-smoke,pytest,bench}
has() { [[ ",$STEPS," == *",$1,"* ]]; }
val() { python -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); k=(d.get('kernel_classes_one_eval') or {}).get('conv3x3_wide') or {}
print('$2', round(d['value'],4), d.get('unit','utt/s'), round(d['ms_per_step'],1), 'ms/step; dominant class', k.get('ms'), 'ms', k.get('tflops'), 'TFLOP/s; frac', (d.get('roofline') or {}).get('frac'), 'power', (d.get('power') or {}).get('mean_w'))" 2>/dev/null || echo "$2 FAILED"; }
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt; nproc >> $O/gpu.txt
if has smoke; then echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 | tee $O/${TAG}_smoke.txt; fi
if has pytest; then
  echo "== pytest -m gpu ${PYTEST_K:+-k \"$PYTEST_K\"}"
  timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=${PYTEST_DURATIONS:-15} ${PYTEST_X:+-x} ${PYTEST_K:+-k "$PYTEST_K"} > $O/${TAG}_pytest_gpu${PYTEST_NAME:+_$PYTEST_NAME}.log 2>&1
  echo "pytest rc=$?"; tail -3 $O/${TAG}_pytest_gpu${PYTEST_NAME:+_$PYTEST_NAME}.log | cut -c1-300
  grep -E "^(FAILED|ERROR)|Memory access|Error" $O/${TAG}_pytest_gpu${PYTEST_NAME:+_$PYTEST_NAME}.log | head -20
  grep -E "^conv_wino|^conv_split" $O/${TAG}_pytest_gpu${PYTEST_NAME:+_$PYTEST_NAME}.log | head -30
  grep -A ${PYTEST_DURATIONS:-15} "slowest" $O/${TAG}_pytest_gpu${PYTEST_NAME:+_$PYTEST_NAME}.log | tee $O/${TAG}_pytest_gpu_durations.txt | head -20
fi
if has micro; then
  echo "== kernel micro-benchmark (VARIANTS=$VARIANTS SHAPES=$SHAPES FUSED=$FUSED)"
  OUT=${TAG}_conv_microbench.json timeout 600 python tools/conv_microbench.py 2>&1 | grep -v amdgpu | tee $O/${TAG}_conv_microbench.txt
fi
if has trace; then
  # phase time stamps of the traced kernel instantiations (bench_conv variant bit 18 = ablation 64): TRACE_VARIANTS, TRACE_SHAPE = "B Cin Cout H W"
  for v in ${TRACE_VARIANTS:-$((1152 + (64 << 12)))}; do
    echo "== phase trace, variant $v, shape ${TRACE_SHAPE:-8 128 128 256 512}"
    SGMSE_TRACE_OUT=$O/trace_$v.bin timeout 120 python -c "
import sys; sys.path.insert(0, '.')
from sgmse_amd import _lib
_lib.load_library(); ctx = _lib.Context('cuda')
B, ci, co, H, W = map(int, '${TRACE_SHAPE:-8 128 128 256 512}'.split())
print('ms per launch', ctx.bench_conv(3, B, ci, co, H, W, variant=$v, iters=3, fused=True))" 2>&1 | grep -v amdgpu
    python tools/analyze_trace.py $O/trace_$v.bin ${TRACE_KIND:-wino} 2>&1 | tee $O/${TAG}_trace_$v.txt | head -${TRACE_LINES:-30}
    rm -f $O/trace_$v.bin
  done
fi
if has power; then
  echo "== power probe (CASES=$CASES)"
  timeout 600 python tools/power_probe.py 2>&1 | grep -v amdgpu | tee $O/${TAG}_power_probe.txt | cut -c1-400
fi
if has ab; then
  echo "== bench A/B"
  : > $O/${TAG}_ab.txt
  IFS=';' read -ra ENTRIES <<< "$AB"
  for ent in "${ENTRIES[@]}"; do
    IFS='|' read -r name args envs <<< "$ent"
    env $envs timeout 400 python bench.py $args > $O/ab_$name.json 2>$O/ab_$name.err
    val $O/ab_$name.json "$name [$envs] [$args]" | tee -a $O/${TAG}_ab.txt
  done
fi
if has bench; then echo "== bench (driver defaults)"; timeout 900 python bench.py > $O/${TAG}_bench_b32.json 2>$O/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-900 $O/${TAG}_bench_b32.json; tail -2 $O/${TAG}_bench.err; fi
if has rocprof; then
  echo "== rocprofv3 kernel trace of the bench command"
  rm -rf $O/prof
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-others > $O/${TAG}_prof_bench.log 2>$O/${TAG}_prof.err; echo "rocprof rc=$?"
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_bench_b32_kernel_stats.csv && head -8 "$f" | cut -c1-200
  rm -rf $O/prof
fi
if has pmc; then
  rm -rf $O/pmc; mkdir -p $O/pmc
  # counters of the dominant kernel on its micro-benchmark, one pass per counter group (the guide's HBM recipe: FETCH_SIZE and
  # WRITE_SIZE in separate passes, calibrated on a float4 stream of known size in the same run)
  echo "== PMC passes (PMC_VARIANT=${PMC_VARIANT:-1152})"
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1)); n=$(echo $grp | cut -d' ' -f1)
    VARIANTS=${PMC_VARIANTS:-128,1152} SHAPES=${PMC_SHAPES:-0} FUSED=1 ROUNDS=1 CALIB=1 OUT=${TAG}_pmc_microbench.json timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc/$n -o pmc -- python tools/conv_microbench.py > $O/pmc/$n.log 2>&1
    echo "pmc pass $i [$grp] rc=$?"
  done
  python tools/summarize_pmc.py $O/pmc/SQ_VALU_MFMA_BUSY_CYCLES $O/pmc/SQ_INSTS_VALU $O/pmc/SQ_LDS_BANK_CONFLICT > $O/${TAG}_pmc_conv.json 2>$O/pmc/summ.err
  python tools/summarize_hbm.py $O/pmc/FETCH_SIZE $O/pmc/WRITE_SIZE $O/${TAG}_pmc_microbench.json > $O/${TAG}_hbm_traffic.json 2>>$O/pmc/summ.err
  python - <<PY
import json
for f in ("$O/${TAG}_pmc_conv.json", "$O/${TAG}_hbm_traffic.json"):
    try:
        d = json.load(open(f))
        for k, v in d.items():
            if "conv3x3" in k: print(f.split("/")[-1], k[:90], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "counters_mean"})
    except Exception as e: print(f, "FAILED", e)
PY
  rm -rf $O/pmc/*/ 2>/dev/null
fi
if has pmcbench; then
  # counters of the bench command's OWN launches at the benched batch: one rocprofv3 --pmc pass per counter group
  # (--kernel-trace only beside it), filtered to the Winograd kernel; PMCB_ARGS adds bench flags (e.g. --no-graph when the replayed
  # graph does not survive the counters); the per-launch listing of the same batch gives the algorithmic bytes
  rm -rf $O/pmcb; mkdir -p $O/pmcb
  PB=${PMCB_BATCH:-32}
  echo "== PMC passes over bench.py --batch $PB (PMCB_ARGS=$PMCB_ARGS)"
  # (visit r05a: at --steps 1 --warmup 1 with the full N = 30 every pass ran into its 420 s limit -- ~170 000 intercepted dispatches per
  #  step -- and the FETCH_SIZE pass ended in a segmentation fault inside the tool.  The launches of a step are the same 2 evaluations
  #  repeated N times, so the passes run the bench command with --N 2: the same kernels at the same batch, 1/15 of the dispatches.)
  PMCB_BASE=${PMCB_BASE:---steps 1 --warmup 0 --N 2 --no-graph}
  if [ -n "$PMCB_GRAPH_PROBE" ]; then      # does counter collection survive the REPLAYED GRAPH? one pass, recorded next to the eager ones
    timeout ${PMCB_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "${PMCB_REGEX:-conv3x3_wino_kernel}" --output-format csv -d $O/pmcb/graph_probe -o pmc -- \
      python bench.py --batch $PB --steps 1 --warmup 0 --N 2 --no-others --no-cpu-baseline --no-profile > $O/pmcb/graph_probe.log 2>&1
    echo "pmcbench graph probe rc=$?; rows: $(find $O/pmcb/graph_probe -name "*counter_collection.csv" -exec cat {} + 2>/dev/null | wc -l); bench line: $(grep -o '"value": [0-9.]*' $O/pmcb/graph_probe.log | head -1)" | tee $O/${TAG}_pmc_bench_graph_probe.txt
  fi
  # round 6: the passes also keep attn_core_kernel (MFMA utilisation of attention) and the known-size calibration streams that
  # SGMSE_PMC_CALIB=1 puts in front of the run (calibrated FETCH_SIZE / WRITE_SIZE: tools/summarize_pmc_bench.py)
  PMCB_CMD="bench.py --batch $PB $PMCB_BASE --no-others --no-cpu-baseline --no-profile $PMCB_ARGS"
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    n=$(echo $grp | cut -d' ' -f1)
    SGMSE_PMC_CALIB=1 timeout ${PMCB_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "${PMCB_REGEX:-conv3x3_wino|attn_core_kernel|calib_stream_kernel}" --output-format csv -d $O/pmcb/$n -o pmc -- \
      python $PMCB_CMD > $O/pmcb/$n.log 2>&1
    rc=$?; echo "pmcbench pass [$grp] rc=$rc; rows: $(find $O/pmcb/$n -name "*counter_collection.csv" -exec cat {} + 2>/dev/null | wc -l); bench line: $(grep -o '"value": [0-9.]*' $O/pmcb/$n.log | head -1)"
    [ $rc -ne 0 ] && tail -5 $O/pmcb/$n.log | cut -c1-300
  done
  SGMSE_PROFILE_DUMP=1 timeout 300 python bench.py --batch $PB --N 2 --steps 1 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2> $O/pmcb/dump.txt
  SRC_HASH=$(python -c "import bench; print(bench.source_hash())" 2>/dev/null)
  python tools/summarize_pmc_bench.py $O/pmcb/SQ_VALU_MFMA_BUSY_CYCLES $O/pmcb/SQ_LDS_BANK_CONFLICT $O/pmcb/FETCH_SIZE $O/pmcb/WRITE_SIZE --dump $O/pmcb/dump.txt \
    --meta "command=$PMCB_CMD" workload=${PMCB_WORKLOAD:-pc16k} batch=$PB "commit=${COMMIT:-unknown}" "source_hash=$SRC_HASH" > $O/${TAG}_pmc_bench_b${PB}.json 2>$O/pmcb/summ.err
  python - <<PY
import json
try:
    d = json.load(open("$O/${TAG}_pmc_bench_b${PB}.json"))
    for k, v in d.items(): print(k[:70], {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != "counters_mean"} if isinstance(v, dict) else v)
except Exception as e: print("pmcbench summary FAILED", e)
PY
  rm -rf $O/pmcb/*/ 2>/dev/null
fi
if has dumps; then
  echo "== per-launch timings of one evaluation (batch 32 and batch 1)"
  for b in ${DUMP_BATCHES:-32 1}; do
    SGMSE_PROFILE_DUMP=1 timeout 300 python bench.py --batch $b --N 2 --steps 1 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2> $O/${TAG}_prof_dump_b${b}.txt
    grep -c sgmse-prof $O/${TAG}_prof_dump_b${b}.txt
  done
fi
if has ragged; then echo "== ragged bench"; timeout 600 python tools/ragged_bench.py --profile > $O/${TAG}_ragged_bench.txt 2>&1; tail -4 $O/${TAG}_ragged_bench.txt | cut -c1-1200; fi
if has dirjob; then echo "== directory job"; timeout 600 python tools/dir_job_bench.py > $O/${TAG}_dir_job.txt 2>&1; tail -5 $O/${TAG}_dir_job.txt | cut -c1-500; fi
if has lengths; then
  echo "== utterance lengths"
  : > $O/${TAG}_length_sweep.txt
  for s in ${LENGTHS:-3.0 3.5 4.0 4.5 5.0}; do
    timeout 300 python bench.py --seconds $s --steps 2 --warmup 1 --no-others --no-cpu-baseline > $O/len_$s.json 2>/dev/null
    val $O/len_$s.json "seconds=$s" | tee -a $O/${TAG}_length_sweep.txt
  done
fi
if has libab; then
  LIB_A=${LIB_A:-sgmse_amd/libsgmse_hip.so}; LIB_B=${LIB_B:?libab needs LIB_B=<path of the second library>}
  : > $O/${TAG}_libab.txt
  runbench() { python -c "
import sys, runpy
from sgmse_amd import _lib
_lib.load_library('$PWD/$1')
sys.argv = ['bench.py'] + '$2'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('   ', round(d['value'], 4), d['unit'], round(d['ms_per_step'], 1), 'ms/step')"; }
  for L in $LIB_A $LIB_B $LIB_B $LIB_A; do
    echo "== lib: $L" | tee -a $O/${TAG}_libab.txt
    [ -z "$LIBAB_SKIP_MICRO" ] && SGMSE_LIB_PATH=$PWD/$L VARIANTS=${VARIANTS:-1152} SHAPES=${SHAPES:-0,1,2} FUSED=1 ROUNDS=3 timeout 200 python tools/conv_microbench.py 2>&1 | grep "ms " | tee -a $O/${TAG}_libab.txt
    runbench $L "--steps 2 --warmup 1 --no-others --no-cpu-baseline --no-profile" | tee -a $O/${TAG}_libab.txt
    [ -z "$LIBAB_SKIP_B1" ] && runbench $L "--batch 1 --steps 3 --warmup 1 --no-others --no-cpu-baseline --no-profile" | tee -a $O/${TAG}_libab.txt
  done
fi
echo "== done"
