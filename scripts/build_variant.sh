#!/bin/bash
# Ablation build: recompiles ONE source of the library with extra -D flags and links a side copy.
# usage: scripts/build_variant.sh <name> <source.hip> [-DFLAG ...]   -> build/mb/libsr_<name>.so
set -e
name=$1; src=$2; shift 2
root=$(cd $(dirname $0)/.. && pwd)
obj=$root/stylerenderer_amd/csrc/_obj
mkdir -p $root/build/mb
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fhip-fp32-correctly-rounded-divide-sqrt "$@" \
  -c $root/stylerenderer_amd/csrc/$src -o $root/build/mb/${name}_${src%.hip}.o
others=$(ls $obj/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/mb/libsr_$name.so $others $root/build/mb/${name}_${src%.hip}.o
echo built build/mb/libsr_$name.so
