#!/bin/bash
# GPU box: cost and effect of the floor-bias cancellation (sign-alternating accumulation) of the split-bf16 data-path
# kernels: build/mb/libsr_{noflip,flip2,flip4}.so (scripts/build_variant.sh ... -DSR_ABL_NOFLIP / -DSR_FLIP_LOG2=1|2)
# against the shipped library (a sign block = one chunk).  Two rounds, alternating.
for round in 1 2; do
for v in "" noflip ${FLIP_VARIANTS:-flip2 flip4}; do
  if [ -z "$v" ]; then unset STYLERENDERER_AMD_LIB; else export STYLERENDERER_AMD_LIB=$PWD/build/mb/libsr_$v.so; fi
  echo "== variant ${v:-flip1 (shipped)}"
  python scripts/bench_split_bf16.py convonly 2>/dev/null | grep "\*\*"
  python scripts/bench_split_bf16.py tonly 2>/dev/null | grep "\*\*" | head -2
  [ $round = 1 ] && python scripts/split_bias_probe.py 2>/dev/null | grep positive | sed -e 's/exact:.*| //'
done
done
