#!/bin/bash
# which PART of conv3x3_split_kernel<SplitH2> disturbs another process's packed FMAs?  (ABLATION=1 build; victim: tools/probes/pk_probe)
for abl in ${ABLS:-0 59 8 3 4 1 2}; do
  (SGMSE_LIB_PATH=$PWD/sgmse_amd/libsgmse_hip_abl.so python - <<PY > /dev/null 2>&1 &
import os, sys, time
sys.path.insert(0, ".")
from sgmse_amd import _lib
_lib.load_library(os.environ["SGMSE_LIB_PATH"]); ctx = _lib.Context("cuda")
t0 = time.time()
while time.time() - t0 < 14:
    ctx.bench_conv(3, 8, 128, 128, 256, 512, variant=128 + ($abl << 12), iters=200, fused=True)
PY
  )
  sleep 9
  r=$(./tools/probes/pk_probe | grep "v_pk_fma_f32" | sed 's/; differing.*//')
  echo "aggressor = split kernel, ablation $abl  ->  victim $r"
  wait; sleep 7
done
