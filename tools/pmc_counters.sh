#!/bin/bash
# Mean per-dispatch value of hardware counters per kernel: one rocprofv3 --pmc pass per argument (each argument is a
# quoted, space-separated counter set that fits one pass).
# usage: tools/pmc_counters.sh <tag> "<bench args>" "<set 1>" ["<set 2>" ...]
TAG=$1; BARGS=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
I=0
for SET in "$@"; do
  I=$((I+1))
  rocprofv3 --pmc $SET --output-format csv -d gpurun_out/pmcc_${TAG}_$I -o p -- python3 bench.py --no-cpu-baseline --no-layer --no-check --steps 2 --warmup 1 $BARGS > gpurun_out/pmcc_${TAG}_$I.log 2>&1 || { tail -5 gpurun_out/pmcc_${TAG}_$I.log; exit 1; }
done
python3 - "$TAG" <<'PY'
import csv, glob, json, re, sys
tag = sys.argv[1]
acc = {}
for f in glob.glob("gpurun_out/pmcc_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0]
        if not (k.startswith("dau::gather") or "dense_gather" in k or "wg_gemm" in k or "split_gather" in k):
            continue
        d = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
        d[0] += float(r["Counter_Value"]); d[1] += 1
out = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open("gpurun_out/%s_pmc_counters.json" % tag, "w"), indent=1)
for k, cs in out.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s %.4g" % (c, v))
PY
