#!/bin/bash
# round 3, visit F: evidence at the current commit -- full GPU suite with its printed error figures, the bench line with driver
# defaults (headline + other_workloads + roofline + cpu_baseline), the rocprofv3 kernel trace of the bench command
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
STEPS=${STEPS:-smoke,pytest,bench,rocprof}
has() { [[ ",$STEPS," == *",$1,"* ]]; }
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt; nproc >> $O/gpu.txt
if has smoke; then echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 | tee $O/r03f_smoke.txt; fi
if has pytest; then echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider ${PYTEST_K:+-k "$PYTEST_K"} > $O/r03f_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r03f_pytest_gpu.log; fi
if has bench; then echo "== bench (driver defaults)"; timeout 900 python bench.py > $O/r03f_bench.json 2>$O/r03f_bench.err; echo "bench rc=$?"; cut -c1-2500 $O/r03f_bench.json; tail -2 $O/r03f_bench.err; fi
if has rocprof; then
  echo "== rocprofv3 kernel trace of the bench command"
  rm -rf $O/prof
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-others > $O/r03f_prof_bench.log 2>$O/r03f_prof.err; echo "rocprof rc=$?"
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r03f_bench_kernel_stats.csv && head -12 "$f" | cut -c1-220
  find $O/prof -name "*kernel_trace.csv" -size +30M -delete
fi
if has b1; then
  echo "== batch 1"
  for i in 1 2; do timeout 300 python bench.py --batch 1 --steps 8 --warmup 2 --no-cpu-baseline --no-others --no-profile 2>/dev/null | tail -1 > $O/r03f_b1_$i.json
    python -c "
import json
d=json.loads(open('$O/r03f_b1_$i.json').read()); print('batch 1 run $i', round(d['value'],4), 'utt/s', round(d['ms_per_step'],1), 'ms per utterance')"; done
fi
if has ragged; then echo "== ragged bench"; timeout 900 python tools/ragged_bench.py --profile > $O/r03f_ragged_bench.txt 2>&1; tail -4 $O/r03f_ragged_bench.txt | cut -c1-1500; fi
if has dirjob; then echo "== directory job"; timeout 900 python tools/dir_job_bench.py > $O/r03f_dir_job.txt 2>&1; tail -5 $O/r03f_dir_job.txt | cut -c1-600; fi
if has dumps; then
  echo "== per-launch timings of one evaluation (batch 32 and batch 1)"
  for b in 32 1; do
    SGMSE_PROFILE_DUMP=1 timeout 600 python bench.py --batch $b --N 2 --steps 1 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2> $O/r03f_prof_dump_b$b.txt
    grep -c sgmse-prof $O/r03f_prof_dump_b$b.txt
  done
fi
