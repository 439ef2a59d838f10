#!/bin/bash
# GPU box: fused transposed-convolution timings over side builds of conv_mfma.hip (scripts/build_variant.sh <name> conv_mfma.hip -D...)
for v in "" "$@"; do
  if [ -z "$v" ]; then unset STYLERENDERER_AMD_LIB; else export STYLERENDERER_AMD_LIB=$PWD/build/mb/libsr_$v.so; fi
  echo "== variant ${v:-base}"; SR_CONVT_FUSED=1 python scripts/bench_convt.py child 2>&1 | grep "convT"
done
