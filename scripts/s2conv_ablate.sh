#!/bin/bash
# GPU box: split-bf16 stride-2 convolution timings over ablation builds of conv_s2_bf16x3.hip (results of the ablated
# builds are wrong by construction; only their time is read)
for v in "" "$@"; do
  if [ -z "$v" ]; then unset STYLERENDERER_AMD_LIB; else export STYLERENDERER_AMD_LIB=$PWD/build/mb/libsr_$v.so; fi
  echo "== variant ${v:-base}"; python scripts/bench_split_bf16.py convonly 2>/dev/null | grep "\*\*"
done
