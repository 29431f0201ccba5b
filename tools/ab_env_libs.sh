#!/bin/bash
# tools/ab_env_libs.sh "<ENV=val ...|-> lib" ... : interleaved bench rounds, each argument = environment settings (or "-") and a
# library path (tuning builds read the knobs); AB_ARGS / AB_STEPS / AB_ROUNDS as in ab_multi.sh
for r in $(seq ${AB_ROUNDS:-2}); do
  for A in "$@"; do
    E=${A% *}; L=${A##* }; [ "$E" = "-" ] && E="X_=1"
    env $E DAU_CONV_LIB=$PWD/$L timeout -k 10 300 python bench.py ${AB_STEPS:---steps 10 --warmup 3} --no-cpu-baseline $AB_ARGS 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$A]', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})"
  done
done
