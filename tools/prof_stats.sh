#!/bin/bash
# rocprofv3 per-kernel summary of one bench.py run; writes gpurun_out/<tag>_kernel_stats.csv
# usage (on the GPU box, from the repo root): tools/prof_stats.sh <tag> [bench args...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/prof_$TAG.log 2>&1 || { tail -5 gpurun_out/prof_$TAG.log; exit 1; }
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$F" gpurun_out/${TAG}_kernel_stats.csv
python3 - "$F" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-72s calls %5s avg_us %10.1f  %5s%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -1 gpurun_out/prof_$TAG.log | cut -c1-300
