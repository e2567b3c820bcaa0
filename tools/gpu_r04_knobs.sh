#!/bin/bash
# First measurement of the next round: the two mechanisms round 3 built after its GPU time was spent (both bit-identical on the
# emulator, both off by default) -- the XCD-aware tile order of the convolutions (SGMSE_CONV_XCD_MAP) and ragged launches over the tile
# columns that exist (SGMSE_RAGGED_PREFIX).  One box, ~6 minutes.  Flip the defaults in engine.h::read_knobs according to the numbers.
set +e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
val() { python -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); k=d['kernel_classes_one_eval']['conv3x3_wide']
print('$2', round(d['value'],4), 'utt/s', round(d['ms_per_step'],1), 'ms/step, dominant class', k['ms'], 'ms', round(k['tflops'],1), 'TFLOP/s')" 2>/dev/null || echo "$2 FAILED"; }
echo "== parity of the two mechanisms on the GPU"
SGMSE_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider -k "xcd or tiles_that_exist" > $O/r04_knobs_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/r04_knobs_pytest.log
: > $O/r04_knobs_ab.txt
run() { local name="$1" args="$2"; shift 2; env "$@" timeout 200 python bench.py $args --steps 2 --warmup 1 --no-others --no-cpu-baseline > $O/k_$name.json 2>/dev/null; val $O/k_$name.json "$name" | tee -a $O/r04_knobs_ab.txt; }
echo "== XCD-aware tile order: T = 512 (16 tiles per row) and T = 448 (14)"
for rep in 1 2; do
  run t512_plain_$rep "" SGMSE_CONV_XCD_MAP=0
  run t512_xcd_$rep "" SGMSE_CONV_XCD_MAP=1
done
run t448_plain "--seconds 3.5" SGMSE_CONV_XCD_MAP=0
run t448_xcd "--seconds 3.5" SGMSE_CONV_XCD_MAP=1
echo "== ragged batch of 32 utterances of 2 ... 6 s: widest-utterance grid vs existing tiles (vs + XCD order); uniform batch of the mean length for scale"
for cfg in "0 0" "1 0" "1 1"; do set -- $cfg
  SGMSE_RAGGED_PREFIX=$1 SGMSE_CONV_XCD_MAP=$2 timeout 300 python tools/ragged_bench.py --modes ragged,uniform 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('prefix $1 xcd $2: ragged', d['ragged_s'], 's  uniform', d['uniform_s'], 's  per frame', d['ragged_vs_uniform_per_frame'])" | tee -a $O/r04_knobs_ab.txt
done
