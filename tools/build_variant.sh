#!/bin/bash
# tools/build_variant.sh <name> <source.hip> "<extra flags>": build/<name>/libdau_conv_hip.so = the objects of the TUNING build
# (-DDAU_TUNING: environment knobs readable) with one source recompiled under extra flags (timing experiments; A/B with
# tools/ab_multi.sh in one gpurun call)
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/../dau-convnet_amd/csrc"
make -s -j8 tuning >/dev/null
mkdir -p ../../build/$name
/opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -I../../include -I. -fvisibility=hidden -mllvm -pragma-unroll-threshold=1000000 -DDAU_TUNING $flags -c $src -o ../../build/$name/variant.o
objs=""
for o in dau_conv_api k_filters k_units k_direct k_gather_mfma k_gather_dot k_dense_bf16 k_dense_wgrad k_dense_split; do
  if [ "$o.hip" = "$src" ]; then objs="$objs ../../build/$name/variant.o"; else objs="$objs tuning_$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/$name/libdau_conv_hip.so $objs tuning_r3_k_dense_bf16.o tuning_r3_k_dense_wgrad.o tuning_s2_k_dense_split.o tuning_s4_k_dense_split.o
echo built build/$name
