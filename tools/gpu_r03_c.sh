#!/bin/bash
# round 3, visit C: after thin-shape loop / 4-row GroupNorm partials / XCD-aware FIR tiles: targeted parity, HBM traffic of FIR,
# bench classes, per-launch dump, ragged bench
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider -k "conv or groupnorm or fir or forward or tile_shape or split or ragged or batch_independence or batch_of_four or poison or memory_held or full_size" > $O/r03c_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r03c_pytest.log
echo "== HBM traffic passes"
rm -rf $O/hbm
export VARIANTS=128 ROUNDS=1 SHAPES=0 FUSED=1 CALIB=1 OUT=hbm_microbench.json
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/hbm/fetch -o p -- python tools/conv_microbench.py > $O/hbm_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/hbm/write -o p -- python tools/conv_microbench.py > $O/hbm_write.log 2>&1; echo "write rc=$?"
unset VARIANTS ROUNDS SHAPES FUSED CALIB OUT
python tools/summarize_hbm.py $O/hbm/fetch $O/hbm/write $O/hbm_microbench.json > $O/r03c_hbm_traffic.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03c_hbm_traffic.json"))
for k, v in d.items():
    if isinstance(v, dict) and "hbm_bytes_per_launch" in v and any(n in k for n in ("conv3x3_split", "fir_", "gn_")):
        print(k[:80], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
PY
echo "== bench"
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-others > $O/r03c_bench.json 2>$O/r03c_bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('$O/r03c_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac']); [print(k,v) for k,v in d['kernel_classes_one_eval'].items()]"
echo "== per-launch dumps"
for b in 32 1; do
  SGMSE_PROFILE_DUMP=1 timeout 600 python bench.py --batch $b --N 2 --steps 1 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2> $O/r03c_prof_dump_b$b.txt
  grep -c sgmse-prof $O/r03c_prof_dump_b$b.txt
done
echo "== batch 1"
timeout 300 python bench.py --batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-others --no-profile > $O/r03c_b1.json 2>/dev/null; python -c "
import json
d=json.loads(open('$O/r03c_b1.json').read().strip().splitlines()[-1]); print('batch 1', d['value'], d['ms_per_step'])"
echo "== ragged bench"
timeout 900 python tools/ragged_bench.py > $O/r03c_ragged_bench.txt 2>&1; tail -5 $O/r03c_ragged_bench.txt
