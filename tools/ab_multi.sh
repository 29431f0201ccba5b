#!/bin/bash
# tools/ab_multi.sh [ENV=VAL ...] lib1 lib2 ... : one bench round per library on the same device
for L in "$@"; do
  DAU_CONV_LIB=$L timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-2], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})"
done
