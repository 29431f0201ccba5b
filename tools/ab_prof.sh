#!/bin/bash
# tools/ab_prof.sh <kernel-name-substring> lib1 lib2 ...: rocprofv3 average of one kernel (NS step) per library, same device
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for L in "$@"; do
  tag=$(basename $(dirname $L))
  DAU_CONV_LIB=$GRAFT_REPO_ROOT/$L rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abp_$tag -o x -- python3 bench.py --no-cpu-baseline --no-layer --steps 10 --warmup 3 $AB_ARGS > gpurun_out/abp_$tag.log 2>&1 || { tail -3 gpurun_out/abp_$tag.log; continue; }
  F=$(find gpurun_out/abp_$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$F" "$PAT" "$tag" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]: print(sys.argv[3], r["Name"][:50], r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
  rm -rf gpurun_out/abp_$tag
done
