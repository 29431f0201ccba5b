#!/bin/bash
# HBM traffic per kernel launch from PMC counters: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), no
# trace domains, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Writes gpurun_out/<tag>_pmc_traffic.json.
# usage (on the GPU box, from the repo root): tools/pmc_traffic.sh <tag> [bench args...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_${TAG}_$C -o $C -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > gpurun_out/pmc_${TAG}_$C.log 2>&1 || { tail -5 gpurun_out/pmc_${TAG}_$C.log; exit 1; }
done
python3 - "$TAG" "$*" <<'PY'
import csv, glob, json, re, sys
tag, args = sys.argv[1], sys.argv[2]
acc = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_%s_%s/**/*counter_collection.csv" % (tag, cname), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != cname:
                continue
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            d = acc.setdefault(k, {}).setdefault(cname, [0.0, 0])
            d[0] += float(r["Counter_Value"]); d[1] += 1
out = {}
for k, v in acc.items():
    if not k.startswith("dau::"):
        continue
    fetch = v.get("FETCH_SIZE", [0.0, 1]); write = v.get("WRITE_SIZE", [0.0, 1])
    fb = fetch[0] / max(fetch[1], 1) * 1024.0      # counter unit: KiB
    wb = write[0] / max(write[1], 1) * 1024.0
    out[k] = dict(fetch_size_bytes_raw=fb, write_size_bytes=wb, hbm_bytes=2.0 * fb + wb, launches=fetch[1])
doc = {"_comment": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes, no trace domains) of `python3 bench.py "
       "--steps 3 --warmup 1 --no-cpu-baseline %s`, MI355X. Counter unit = KiB per dispatch (mean over launches). hbm_bytes "
       "applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide coalesced read): "
       "2*FETCH + WRITE." % args, "kernels": out}
json.dump(doc, open("gpurun_out/%s_pmc_traffic.json" % tag, "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes"])[:8]:
    print("%-60s %8.3f GB/launch" % (k[:60], v["hbm_bytes"] / 1e9))
PY
