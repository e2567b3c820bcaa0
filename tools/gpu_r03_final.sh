#!/bin/bash
# round 3, final visit: evidence at the final commit -- smoke, the full GPU suite with its printed error figures, the bench line with
# driver defaults (headline + other_workloads + roofline + cpu_baseline), the rocprofv3 kernel trace of the bench command, the
# experimental arrival mode of the GroupNorm tails, the Schroedinger-bridge probe, ragged / directory-job rates, per-launch dumps
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
STEPS=${STEPS:-smoke,pytest,bench,rocprof,tailmode,sb,ragged,dirjob,dumps}
has() { [[ ",$STEPS," == *",$1,"* ]]; }
val() { python -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],4), 'utt/s', round(d['ms_per_step'],1), 'ms/step', 'tail jobs', d.get('gn_tail_jobs_per_eval'))" 2>/dev/null || echo "$2 FAILED"; }
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt; nproc >> $O/gpu.txt
if has smoke; then echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 | tee $O/r03_smoke.txt; fi
if has pytest; then echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider ${PYTEST_K:+-k "$PYTEST_K"} > $O/r03_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r03_pytest_gpu.log | cut -c1-300; grep -E "GroupNorm jobs|Memory access" $O/r03_pytest_gpu.log; fi
if has bench; then echo "== bench (driver defaults)"; timeout 900 python bench.py > $O/r03_bench_b32.json 2>$O/r03_bench.err; echo "bench rc=$?"; cut -c1-700 $O/r03_bench_b32.json; tail -2 $O/r03_bench.err; fi
if has rocprof; then
  echo "== rocprofv3 kernel trace of the bench command"
  rm -rf $O/prof
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-others > $O/r03_prof_bench.log 2>$O/r03_prof.err; echo "rocprof rc=$?"
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r03_bench_b32_kernel_stats.csv && head -8 "$f" | cut -c1-200
  rm -rf $O/prof
fi
if has tailmode; then
  echo "== GroupNorm tails with device-coherent accesses and no cache maintenance (SGMSE_GN_TAIL_MODE=1, experimental)"
  : > $O/r03_tail_mode1_ab.txt
  B1="--batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-others --no-profile"
  run() { local name="$1" args="$2"; shift 2; env "$@" timeout 300 python bench.py $args > $O/i_$name.json 2>/dev/null; val $O/i_$name.json "$name" | tee -a $O/r03_tail_mode1_ab.txt; }
  run b1_none "$B1" SGMSE_GN_TAIL=0
  run b1_tail_mode1 "$B1" SGMSE_GN_TAIL=1 SGMSE_GN_TAIL_MODE=1
  run b1_tail_mode1_64k "$B1" SGMSE_GN_TAIL=1 SGMSE_GN_TAIL_MODE=1 SGMSE_GN_TAIL_MAX_PAIRS=65536
  run b1_tail_mode0 "$B1" SGMSE_GN_TAIL=1 SGMSE_GN_TAIL_MODE=0
  run b32_tail_mode1 "--steps 2 --warmup 1 --no-cpu-baseline --no-others --no-profile" SGMSE_GN_TAIL=1 SGMSE_GN_TAIL_MODE=1
fi
if has sb; then echo "== Schroedinger-bridge probe"; timeout 300 python tools/sb_probe.py > $O/r03_sb_probe.txt 2>&1; grep -v amdgpu $O/r03_sb_probe.txt | cut -c1-260; fi
if has ragged; then echo "== ragged bench"; timeout 600 python tools/ragged_bench.py --profile > $O/r03_ragged_bench.txt 2>&1; tail -4 $O/r03_ragged_bench.txt | cut -c1-1200; fi
if has dirjob; then echo "== directory job"; timeout 600 python tools/dir_job_bench.py > $O/r03_dir_job.txt 2>&1; tail -5 $O/r03_dir_job.txt | cut -c1-500; fi
if has dumps; then
  echo "== per-launch timings of one evaluation (batch 32 and batch 1)"
  for b in 32 1; do
    SGMSE_PROFILE_DUMP=1 timeout 300 python bench.py --batch $b --N 2 --steps 1 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2> $O/r03_prof_dump_b${b}_final.txt
    grep -c sgmse-prof $O/r03_prof_dump_b${b}_final.txt
  done
fi
