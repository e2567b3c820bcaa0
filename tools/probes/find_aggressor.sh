#!/bin/bash
# which of the library's kernels, running in ANOTHER process, changes the packed-FMA results of tools/probes/pk_probe?
# (build the victim first: hipcc --offload-arch=gfx950 -O3 -o tools/probes/pk_probe tools/probes/pk_fma_probe.hip)
python tools/probes/ops_under_load_probe.py --list ${MORE:+--more} 2>/dev/null | grep -v amdgpu | grep "${ONLY:-.}" > /tmp/ops.txt
while IFS= read -r op; do
  (LOAD_SECONDS=14 python tools/probes/ops_under_load_probe.py --as-load "$op" > /dev/null 2>&1 &)
  sleep 9
  r=$(./tools/probes/pk_probe | grep "^v_pk_fma_f32" | sed 's/; by lane.*//')
  echo "load = [$op]  ->  $r"
  wait
  sleep 6
done < /tmp/ops.txt
