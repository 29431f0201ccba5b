#!/bin/bash
# interleaved same-device runs of bench.py under different settings: tools/ab_env.sh "<env A>" "<env B>" ... (each quoted: VAR=val VAR2=val)
# the environment knobs exist in the tuning build only (make -C dau-convnet_amd/csrc tuning)
export DAU_CONV_LIB=${DAU_CONV_LIB:-${GRAFT_REPO_ROOT:-$PWD}/dau-convnet_amd/dau_conv/libdau_conv_hip_tuning.so}
R=2
for r in $(seq $R); do
  for E in "$@"; do
    env $E timeout -k 10 300 python bench.py $AB_ARGS ${AB_STEPS:---steps 10 --warmup 3} --no-cpu-baseline 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$E]', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})"
  done
done
