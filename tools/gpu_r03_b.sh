#!/bin/bash
# round 3, visit B: same-box A/B against the round-2 library, batch-1 kernel trace, chunking knobs, new parity tests,
# directory job vs bench rate, full default bench line
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],4), 'utt/s', round(d['ms_per_step'],1), 'ms/step')"; }
echo "== A/B r02 lib vs current (batch 32, 2 steps each, ABAB)" | tee $O/r03b_ab.txt
for i in 1 2; do
  timeout 300 python tools/ab_bench.py tools/ab/libsgmse_hip_r02.so --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-profile > $O/ab_r02_$i.json 2>$O/ab_r02_$i.err; val $O/ab_r02_$i.json "r02 lib run $i" | tee -a $O/r03b_ab.txt
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-profile > $O/ab_cur_$i.json 2>$O/ab_cur_$i.err; val $O/ab_cur_$i.json "current run $i" | tee -a $O/r03b_ab.txt
done
echo "== batch 1: chunking knobs" | tee $O/r03b_b1_knobs.txt
for k in 8 32 128; do
  SGMSE_CHUNK_MAX_TILES=$k timeout 300 python bench.py --batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-others --no-profile > $O/b1_k$k.json 2>/dev/null; val $O/b1_k$k.json "batch 1 CHUNK_MAX_TILES=$k" | tee -a $O/r03b_b1_knobs.txt
done
for k in 32 128; do
  SGMSE_CHUNK_MAX_TILES=$k timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-others --no-profile > $O/b32_k$k.json 2>/dev/null; val $O/b32_k$k.json "batch 32 CHUNK_MAX_TILES=$k" | tee -a $O/r03b_b1_knobs.txt
done
echo "== batch 1 kernel trace"
rm -rf $O/prof_b1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_b1 -o t -- python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-others --no-profile > $O/prof_b1.log 2>&1
python tools/summarize_kernel_trace.py $O/prof_b1 0.6 > $O/r03b_b1_trace_summary.txt 2>&1; head -40 $O/r03b_b1_trace_summary.txt
find $O/prof_b1 -name "*.csv" -size +20M -delete
echo "== new parity tests"
timeout 1200 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider -k "T512 or benched_shape or enhancement_script or ragged or graph_equals or repeated_sampler or philox" > $O/r03b_pytest_new.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r03b_pytest_new.log
echo "== directory job vs bench rate"
timeout 900 python tools/dir_job_bench.py --files 256 > $O/r03b_dir_job.json 2>$O/r03b_dir_job.err; cat $O/r03b_dir_job.json; tail -3 $O/r03b_dir_job.err
echo "== full default bench line"
timeout 900 python bench.py > $O/r03b_bench_default.json 2>$O/r03b_bench_default.err; echo "bench rc=$?"; cut -c1-300 $O/r03b_bench_default.json
