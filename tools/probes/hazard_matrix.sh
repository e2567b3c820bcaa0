#!/bin/bash
# victim instruction forms of tools/probes/pk_probe (built with -fno-slp-vectorize -ffp-contract=off) beside the aggressor
# (conv3x3_split_kernel<SplitH2> looping in a second process) and alone
echo "-- alone"; ./tools/probes/pk_probe
(LOAD_SECONDS=40 python tools/probes/ops_under_load_probe.py --as-load "conv3x3 fp16x2 256->256 @32x8" > /dev/null 2>&1 &)
sleep 12
echo "-- beside conv3x3_split_kernel<SplitH2> of a second process"; ./tools/probes/pk_probe
wait
