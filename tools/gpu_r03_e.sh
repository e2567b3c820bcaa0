#!/bin/bash
# round 3, visit E: batch-1 / batch-32 A/B against the round-2 library on one box, ragged profile, sampler / graph-update tests
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],4), 'utt/s', round(d['ms_per_step'],1), 'ms/step')"; }
timeout 900 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider -k "sampler or schroedinger or ragged or graph or enhancement_script or philox or error_behaviour or weight_reload" > $O/r03e_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r03e_pytest.log; grep -E "sb_" $O/r03e_pytest.log
echo "== A/B batch 1" | tee $O/r03e_ab.txt
for i in 1 2; do
  timeout 300 python tools/ab_bench.py tools/ab/libsgmse_hip_r02.so --batch 1 --steps 8 --warmup 2 --no-cpu-baseline --no-others --no-profile > $O/abe_r02_b1_$i.json 2>/dev/null; val $O/abe_r02_b1_$i.json "r02 lib  batch 1 run $i" | tee -a $O/r03e_ab.txt
  timeout 300 python bench.py --batch 1 --steps 8 --warmup 2 --no-cpu-baseline --no-others --no-profile > $O/abe_cur_b1_$i.json 2>/dev/null; val $O/abe_cur_b1_$i.json "current  batch 1 run $i" | tee -a $O/r03e_ab.txt
done
echo "== A/B batch 32" | tee -a $O/r03e_ab.txt
for i in 1 2; do
  timeout 300 python tools/ab_bench.py tools/ab/libsgmse_hip_r02.so --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-profile > $O/abe_r02_$i.json 2>/dev/null; val $O/abe_r02_$i.json "r02 lib  batch 32 run $i" | tee -a $O/r03e_ab.txt
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-profile > $O/abe_cur_$i.json 2>/dev/null; val $O/abe_cur_$i.json "current  batch 32 run $i" | tee -a $O/r03e_ab.txt
done
echo "== ragged bench with class profile"
timeout 900 python tools/ragged_bench.py --profile > $O/r03e_ragged_bench.txt 2>&1; tail -4 $O/r03e_ragged_bench.txt | cut -c1-1500
