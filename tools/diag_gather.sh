#!/bin/bash
# forward gather time of the north-star shape under the diagnostic builds (results are garbage, timing only)
for L in dau-convnet_amd/dau_conv build/diag_NOLDS build/diag_NOBARRIER build/diag_NOLDSDDAU_DIAG_NOBARRIER; do
  for D in 0 1; do
    DAU_CONV_LIB=$GRAFT_REPO_ROOT/$L/libdau_conv_hip.so DAU_GATHER_DEBUG=$D timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L debug=$D', d['roofline']['kernels']['gather_sum_fwd']['avg_ms'], d['roofline']['kernels']['gather_sum_dx']['avg_ms'])"
  done
done
