#!/bin/bash
# tools/ab_multi.sh lib1 lib2 ... : interleaved bench rounds per library on the same device (AB_ARGS: bench arguments,
# AB_STEPS: "--steps K --warmup W", AB_ROUNDS: rounds, default 2)
for r in $(seq ${AB_ROUNDS:-2}); do
  for L in "$@"; do
    DAU_CONV_LIB=$PWD/$L timeout -k 10 300 python bench.py ${AB_STEPS:---steps 10 --warmup 3} --no-cpu-baseline $AB_ARGS 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-2], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})"
  done
done
