#!/bin/bash
# Sample GPU power / clocks with rocm-smi while bench.py runs: tools/power_trace.sh [bench args]
python bench.py --steps 400 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
N=0
while kill -0 $BP 2>/dev/null; do
  L=$(rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Package Power|sclk|junction" | sed 's/.*: //' | tr '\n' ' ')
  P=$(echo "$L" | grep -o -E "[0-9]+\.[0-9]+ *$" | head -1)
  case "$L" in *"(1"[0-9][0-9][0-9]"Mhz"*|*"(2"[0-9][0-9][0-9]"Mhz"*) N=$((N+1)); [ $N -le 14 ] && echo "$L";; esac
  sleep 0.7
done
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
cut -c1-200 gpurun_out/power_bench.json
