#!/bin/bash
# tools/ab_prof_env.sh <kernel-name-substring> "<env A>" "<env B>" ...: rocprofv3 average of one kernel per setting, same device
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for E in "$@"; do
  i=$((i+1))
  export $E
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abp_$i -o x -- python3 bench.py --no-cpu-baseline --no-layer --steps 10 --warmup 3 $AB_ARGS > gpurun_out/abp_$i.log 2>&1 || { tail -3 gpurun_out/abp_$i.log; }
  for v in $E; do unset ${v%%=*}; done
  F=$(find gpurun_out/abp_$i -name "*kernel_stats.csv" | head -1)
  python3 - "$F" "$PAT" "$E" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]: print("[%s]" % sys.argv[3], r["Name"][:50], r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
  rm -rf gpurun_out/abp_$i
done
