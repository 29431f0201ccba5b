#!/bin/bash
# Same-device A/B of two builds of libdau_conv_hip.so: interleaved rounds in one gpurun call.
# usage: tools/ab_bench.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-2}
for r in $(seq $R); do
  for L in "$A" "$B"; do
    DAU_CONV_LIB=$L timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-2], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})"
  done
done
