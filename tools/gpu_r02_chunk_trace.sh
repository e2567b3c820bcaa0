#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof_chunk
for cfg in "SGMSE_CHUNK_MAX_TILES=32 SGMSE_COARSE_SPLITK_DIV=4" "SGMSE_CHUNK_MAX_TILES=32 SGMSE_COARSE_SPLITK_DIV=1000000"; do
  tag=$(echo $cfg | tr -c 'A-Za-z0-9\n' '_')
  env $cfg timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chunk/$tag -o t -- python bench.py --batch 1 --N 2 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
  f=$(find gpurun_out/prof_chunk/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"; [ -n "$f" ] && head -14 "$f" | cut -c1-200
  find gpurun_out/prof_chunk/$tag -name "*kernel_trace.csv" -size +20M -delete
done
