#!/bin/bash
# GPU box: Winograd forward kernel timings over side builds of conv_wino.hip
# (scripts/build_variant.sh <name> conv_wino.hip -DWINO_RD=... ; see the schedule strings in k_conv_wino)
for v in "" "$@"; do
  if [ -z "$v" ]; then unset STYLERENDERER_AMD_LIB; else export STYLERENDERER_AMD_LIB=$PWD/build/mb/libsr_$v.so; fi
  echo "== variant ${v:-base}"; SR_WINOGRAD=1 python scripts/bench_wino.py child 2>&1 | grep "^wino  B"
done
