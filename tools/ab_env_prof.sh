#!/bin/bash
# per-kernel averages (rocprofv3 --stats) of one bench.py run under each given environment setting:
#   tools/ab_env_prof.sh "<bench args>" "<kernel name substring>" "VAR=a" "VAR=b" ...
# the environment knobs exist in the tuning build only (make -C dau-convnet_amd/csrc tuning)
export DAU_CONV_LIB=${DAU_CONV_LIB:-${GRAFT_REPO_ROOT:-$PWD}/dau-convnet_amd/dau_conv/libdau_conv_hip_tuning.so}
ARGS=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for E in "$@"; do
  i=$((i+1))
  env $E true
  ( export $E; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abp_$i -o r -- python3 bench.py --no-cpu-baseline --no-layer $ARGS > gpurun_out/abp_$i.log 2>&1 )
  F=$(find gpurun_out/abp_$i -name "*kernel_stats.csv" | head -1)
  python3 - "$F" "$PAT" "$E" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print("[%s] %-60s calls %4s avg_us %9.1f" % (sys.argv[3], r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
