#!/bin/bash
# Forward / dx time of the EXACT gather (--no-split) of one workload (DIAG_ARGS, default: the north-star shape) under diagnostic builds of k_gather_mfma.hip (their results are
# garbage, timing only): NOLDS = no tile reads, NOBARRIER = no per-channel barrier, DAU_GATHER_DEBUG=1 = no plane DMA.
# Run on the GPU box from the repo root (builds the variants with hipcc first).
set -e
cd "$GRAFT_REPO_ROOT/dau-convnet_amd/csrc"
make -s -j8 tuning >/dev/null 2>&1
for V in NOLDS NOBARRIER "NOLDS -DDAU_DIAG_NOBARRIER"; do
  T=$(echo $V | tr -d ' -')
  mkdir -p ../../build/diag_$T
  /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -I../../include -I. -fvisibility=hidden -DDAU_TUNING -DDAU_DIAG_$V -c k_gather_mfma.hip -o /tmp/k_gm_$T.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/diag_$T/libdau_conv_hip.so tuning_dau_conv_api.o tuning_k_filters.o tuning_k_units.o tuning_k_direct.o /tmp/k_gm_$T.o tuning_k_gather_dot.o tuning_k_dense_bf16.o tuning_k_dense_wgrad.o tuning_k_dense_split.o tuning_r3_k_dense_bf16.o tuning_r3_k_dense_wgrad.o tuning_s2_k_dense_split.o tuning_s4_k_dense_split.o
done
mkdir -p build/diag_base 2>/dev/null; cd "$GRAFT_REPO_ROOT"; mkdir -p build/diag_base; cp dau-convnet_amd/dau_conv/libdau_conv_hip_tuning.so build/diag_base/libdau_conv_hip.so
for L in build/diag_base build/diag_NOLDS build/diag_NOBARRIER build/diag_NOLDSDDAU_DIAG_NOBARRIER; do
  for D in 0 1; do
    DAU_CONV_LIB=$GRAFT_REPO_ROOT/$L/libdau_conv_hip.so DAU_GATHER_DEBUG=$D timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-layer --no-check --no-split $DIAG_ARGS 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L debug=$D', d['roofline']['kernels']['gather_sum_fwd']['avg_ms'], d['roofline']['kernels']['gather_sum_dx']['avg_ms'])"
  done
done
