#!/bin/bash
# round 3, visit G: GroupNorm coefficients in the convolution tails (conv_gn_tail) -- parity on the GPU, A/B at batch 1 and 32 on one box,
# and the per-level effect of chunked split-K on the 64 x 128 / 128 x 256 levels at batch 1 (existing knobs)
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
val() { python -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],4), 'utt/s', round(d['ms_per_step'],1), 'ms/step', 'tail jobs', d.get('gn_tail_jobs_per_eval'))" 2>/dev/null || echo "$2 FAILED"; }
timeout 900 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider -k "tail or forward_matches or ragged or batch_independence or graph_equals or tile_shape or split_k or memory_held or batch_of_four or baseline_configuration" > $O/r03g_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r03g_pytest.log; grep -E "GroupNorm jobs" $O/r03g_pytest.log
B1="--batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-others --no-profile"
B32="--steps 2 --warmup 1 --no-cpu-baseline --no-others --no-profile"
: > $O/r03g_ab.txt
run() { # name, bench args, env...
  local name="$1" args="$2"; shift 2
  env "$@" timeout 300 python bench.py $args > $O/g_$name.json 2>/dev/null; val $O/g_$name.json "$name" | tee -a $O/r03g_ab.txt
}
echo "== batch 1"
run b1_tail_off "$B1" SGMSE_GN_TAIL=0
run b1_tail_default "$B1" SGMSE_GN_TAIL=1
run b1_tail_64k "$B1" SGMSE_GN_TAIL_MAX_PAIRS=65536
run b1_tail_all "$B1" SGMSE_GN_TAIL_MAX_PAIRS=1000000000
run b1_tail_off_2 "$B1" SGMSE_GN_TAIL=0
run b1_tail_default_2 "$B1" SGMSE_GN_TAIL=1
run b1_chunk32 "$B1" SGMSE_CHUNK_MAX_TILES=32
run b1_chunk128 "$B1" SGMSE_CHUNK_MAX_TILES=128
run b1_chunk32_nofold "$B1" SGMSE_CHUNK_MAX_TILES=32 SGMSE_FOLD_SHORTCUT=0
run b1_chunk128_nofold "$B1" SGMSE_CHUNK_MAX_TILES=128 SGMSE_FOLD_SHORTCUT=0
run b1_tmb1024 "$B1" SGMSE_TILE_MIN_BLOCKS=1024
echo "== batch 32"
run b32_tail_off "$B32" SGMSE_GN_TAIL=0
run b32_tail_default "$B32" SGMSE_GN_TAIL=1
run b32_tail_all "$B32" SGMSE_GN_TAIL_MAX_PAIRS=1000000000
run b32_chunk32 "$B32" SGMSE_CHUNK_MAX_TILES=32
echo "== per-launch dumps"
dump() { local name="$1" b="$2"; shift 2
  env SGMSE_PROFILE_DUMP=1 "$@" timeout 300 python bench.py --batch $b --N 2 --steps 1 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2> $O/r03g_dump_$name.txt; grep -c sgmse-prof $O/r03g_dump_$name.txt; }
dump b1_default 1 SGMSE_GN_TAIL=1
dump b1_chunk128_nofold 1 SGMSE_CHUNK_MAX_TILES=128 SGMSE_FOLD_SHORTCUT=0
dump b32_default 32 SGMSE_GN_TAIL=1
dump b32_chunk128_nofold 32 SGMSE_CHUNK_MAX_TILES=128 SGMSE_FOLD_SHORTCUT=0
