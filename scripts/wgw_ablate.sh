#!/bin/bash
# GPU box: scripts/bench_wgw_ab.py over side builds of conv_wgrad_wino.hip (scripts/build_variant.sh <name> conv_wgrad_wino.hip -D...)
for v in "" "$@"; do
  if [ -z "$v" ]; then unset STYLERENDERER_AMD_LIB; else export STYLERENDERER_AMD_LIB=$PWD/build/mb/libsr_$v.so; fi
  echo "== variant ${v:-base}"; python scripts/bench_wgw_ab.py 2>&1 | grep -v amdgpu.ids | sed -n 7,12p
done
