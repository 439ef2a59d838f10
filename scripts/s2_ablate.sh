#!/bin/bash
# GPU box: stride-2 weight-gradient timings over side builds of conv_wgrad_mfma.hip
for v in "" "$@"; do
  if [ -z "$v" ]; then unset STYLERENDERER_AMD_LIB; else export STYLERENDERER_AMD_LIB=$PWD/build/mb/libsr_$v.so; fi
  echo "== variant ${v:-base}"; python scripts/bench_wgrad_s2.py 2>&1 | grep "^B" | tail -4
done
