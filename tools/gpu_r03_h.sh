#!/bin/bash
# round 3, visit H: arrive-last with separate release / acquire fences -- parity on the GPU, then the four on/off combinations of
# the GroupNorm tails and the fused split-K reduce at batch 1 and batch 32 on one box
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
val() { python -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],4), 'utt/s', round(d['ms_per_step'],1), 'ms/step', 'tail jobs', d.get('gn_tail_jobs_per_eval'))" 2>/dev/null || echo "$2 FAILED"; }
# the ragged full-width test faulted in visit G with both features on: which one, and in which launch
for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg
  SGMSE_GN_TAIL=$1 SGMSE_SPLITK_FUSED=$2 SGMSE_DEBUG_SYNC=1 timeout 300 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider -k "ragged_batch_gives" > $O/r03h_ragged_tail$1_fused$2.log 2>&1
  echo "ragged test, tail=$1 fused=$2: rc=$? $(grep -c 'sgmse-dbg' $O/r03h_ragged_tail$1_fused$2.log) launches traced"; grep -E "Memory access|passed|failed" $O/r03h_ragged_tail$1_fused$2.log | head -3
  grep "sgmse-dbg" $O/r03h_ragged_tail$1_fused$2.log | tail -4 > $O/r03h_ragged_tail$1_fused$2_last.txt; cat $O/r03h_ragged_tail$1_fused$2_last.txt | cut -c1-200
  grep -v "sgmse-dbg" $O/r03h_ragged_tail$1_fused$2.log | tail -30 > $O/r03h_ragged_tail$1_fused$2_tail.txt; rm -f $O/r03h_ragged_tail$1_fused$2.log
done
timeout 900 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider -k "${PYTEST_K:-tail or ragged or batch_independence or graph_equals or tile_shape or split_k or memory_held or batch_of_four}" > $O/r03h_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r03h_pytest.log | cut -c1-300; grep -E "GroupNorm jobs|Memory access" $O/r03h_pytest.log
B1="--batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-others --no-profile"
B32="--steps 2 --warmup 1 --no-cpu-baseline --no-others --no-profile"
: > $O/r03h_ab.txt
run() { local name="$1" args="$2"; shift 2
  env "$@" timeout 300 python bench.py $args > $O/h_$name.json 2>/dev/null; val $O/h_$name.json "$name" | tee -a $O/r03h_ab.txt; }
echo "== batch 1"
for rep in 1; do
run b1_none_$rep "$B1" SGMSE_GN_TAIL=0 SGMSE_SPLITK_FUSED=0
run b1_splitk_$rep "$B1" SGMSE_GN_TAIL=0 SGMSE_SPLITK_FUSED=1
run b1_tail_$rep "$B1" SGMSE_GN_TAIL=1 SGMSE_SPLITK_FUSED=0
run b1_both_$rep "$B1" SGMSE_GN_TAIL=1 SGMSE_SPLITK_FUSED=1
done
run b1_tail_64k "$B1" SGMSE_GN_TAIL=1 SGMSE_SPLITK_FUSED=0 SGMSE_GN_TAIL_MAX_PAIRS=65536
echo "== batch 32"
run b32_none "$B32" SGMSE_GN_TAIL=0 SGMSE_SPLITK_FUSED=0
run b32_tail "$B32" SGMSE_GN_TAIL=1 SGMSE_SPLITK_FUSED=0
run b32_both "$B32" SGMSE_GN_TAIL=1 SGMSE_SPLITK_FUSED=1
echo "== per-launch dumps, batch 1"
dump() { local name="$1" b="$2"; shift 2
  env SGMSE_PROFILE_DUMP=1 "$@" timeout 300 python bench.py --batch $b --N 2 --steps 1 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2> $O/r03h_dump_$name.txt; grep -c sgmse-prof $O/r03h_dump_$name.txt; }
dump b1_none 1 SGMSE_GN_TAIL=0 SGMSE_SPLITK_FUSED=0
dump b1_both 1 SGMSE_GN_TAIL=1 SGMSE_SPLITK_FUSED=1
