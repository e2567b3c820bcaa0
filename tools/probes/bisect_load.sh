for cfg in "" "SGMSE_CONV_SPLIT=0" "SGMSE_WINO=0" "SGMSE_FUSE_GN_STATS=0" "SGMSE_NOFOLD_LEVELS=127" "SGMSE_CONV_XCD_MAP=0" "SGMSE_TILE_MIN_BLOCKS=1"; do
  echo "== [$cfg]"
  env $cfg PROBE_T=64 REPS=30 LOAD_SECONDS=30 python tools/probes/concurrency_bits_probe.py 2>&1 | grep -v amdgpu | grep "under outside\|alone"
done
