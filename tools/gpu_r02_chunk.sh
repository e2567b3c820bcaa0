#!/bin/bash
# Chunked 4-row split kernel on more levels + split-K of it (after the reduce kernel's chunk loop went outside): batch 1 / 32
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for cm in 7 32; do
  SGMSE_CHUNK_MAX_TILES=$cm SGMSE_COARSE_SPLITK_DIV=4 timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "split_k_of_the_coarse or tile_shape or forward_matches_reference_full_width" 2>&1 | tail -1
done
for cfg in "SGMSE_CHUNK_MAX_TILES=7 SGMSE_COARSE_SPLITK_DIV=1000000" "SGMSE_CHUNK_MAX_TILES=7 SGMSE_COARSE_SPLITK_DIV=4" "SGMSE_CHUNK_MAX_TILES=8 SGMSE_COARSE_SPLITK_DIV=4" "SGMSE_CHUNK_MAX_TILES=32 SGMSE_COARSE_SPLITK_DIV=4" "SGMSE_CHUNK_MAX_TILES=32 SGMSE_COARSE_SPLITK_DIV=1"; do
  for b in 1 32; do
    env $cfg timeout 600 python bench.py --batch $b --steps $([ $b = 1 ] && echo 3 || echo 1) --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg batch $b:', round(d['ms_per_step']/1e3,4), 's per step', round(d['value'],3), 'utt/s', {k:(v['ms'],v['launches']) for k,v in d['kernel_classes_one_eval'].items() if k.startswith('conv')})"
  done
done
SGMSE_CHUNK_MAX_TILES=32 SGMSE_COARSE_SPLITK_DIV=4 SGMSE_PROFILE_DUMP=1 timeout 600 python bench.py --batch 1 --N 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_b1_chunk.log 2> gpurun_out/prof_dump_b1_chunk32.txt
grep -c sgmse-prof gpurun_out/prof_dump_b1_chunk32.txt
