#!/bin/bash
# per-kernel TFLOP/s for a list of ad-hoc shapes "N,S,F,H,W,G,k": tools/shape_sweep.sh shape1 shape2 ...
for S in "$@"; do
  timeout -k 10 300 python bench.py --shape $S --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$S', d['ms_per_step'], {k:(v['avg_ms'],v['tflops']) for k,v in d['roofline']['kernels'].items()}, d['roofline']['whole_step_tflops'])" || exit 1
done
