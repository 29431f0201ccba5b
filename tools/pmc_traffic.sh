#!/bin/bash
# HBM traffic per kernel from PMC counters: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), no trace
# domains, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Writes gpurun_out/<tag>_pmc_traffic.json in the format
# bench.py reads from profiles/r4_pmc_traffic.json (merge the "runs" entries there and commit).
# usage (on the GPU box, from the repo root): tools/pmc_traffic.sh <tag> <workload> <io> [more bench args...]
TAG=$1; WL=$2; IO=$3; shift 3
STEPS=3; WARM=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_${TAG}_$C -o $C -- python3 bench.py --no-cpu-baseline --no-layer --no-check --steps $STEPS --warmup $WARM --workload $WL --io $IO "$@" > gpurun_out/pmc_${TAG}_$C.log 2>&1 || { tail -5 gpurun_out/pmc_${TAG}_$C.log; exit 1; }
done
python3 - "$TAG" "$WL" "$IO" "$STEPS" "$WARM" "$*" <<'PY'
import csv, glob, hashlib, json, re, sys
tag, wl, io, steps, warm, extra = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
nsteps = steps + warm
acc = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_%s_%s/**/*counter_collection.csv" % (tag, cname), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != cname:
                continue
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            d = acc.setdefault(k, {}).setdefault(cname, [0.0, 0])
            d[0] += float(r["Counter_Value"]); d[1] += 1
kern, fam = {}, {}
for k, v in acc.items():
    if not k.startswith("dau::"):
        continue
    fetch = v.get("FETCH_SIZE", [0.0, 0]); write = v.get("WRITE_SIZE", [0.0, 0])
    fb, wb = fetch[0] * 1024.0, write[0] * 1024.0          # counter unit: KiB; totals over all dispatches of the run
    per_step = (2.0 * fb + wb) / nsteps                       # gfx950: FETCH_SIZE reports half of a wide coalesced read
    kern[k] = dict(dispatches_per_step=fetch[1] / float(nsteps), fetch_size_bytes_raw_per_step=fb / nsteps,
                   write_size_bytes_per_step=wb / nsteps, hbm_bytes_per_step=per_step)
    family = k.split("<")[0]
    fam[family] = fam.get(family, 0.0) + per_step
# per PASS of the dominant kernels: the step runs two gather-sum passes (forward, dx) and one gather-dot pass
# (a guarded member that does not run is dispatched too: a few KB; the family a call really takes carries the bytes)
for family, passes in (("dau::gather_mfma_kernel", 2), ("dau::gather_dot_kernel", 1), ("dau::s2::split_gather_kernel", 2),
                       ("dau::s3::split_gather_kernel", 2), ("dau::s4::split_gather_kernel", 2)):
    if family in fam:
        kern[family] = dict(hbm_bytes_per_pass=fam[family] / passes, passes_per_step=passes)
sys.path.insert(0, ".")
import bench
lib = bench.lib_fingerprint()          # the library's build id = fingerprint of its sources (path independent), see bench.py
doc = {"_comment": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes, no trace domains) of `python3 bench.py "
       "--steps %d --warmup %d --no-cpu-baseline --no-layer --workload %s --io %s %s`, MI355X. Counter unit KiB per dispatch, summed "
       "over the run and divided by its %d steps. hbm bytes apply the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE "
       "reports half of a wide coalesced read): 2*FETCH + WRITE." % (steps, warm, wl, io, extra, nsteps),
       "runs": {"%s/%s" % (wl, io): {"src_sha256_16": lib, "kernels": kern,
                                      "step_total_hbm_bytes": sum(v["hbm_bytes_per_step"] for v in kern.values() if "hbm_bytes_per_step" in v)}}}
json.dump(doc, open("gpurun_out/%s_pmc_traffic.json" % tag, "w"), indent=1)
for k, v in sorted(kern.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_step", 0))[:9]:
    if "hbm_bytes_per_step" in v:
        print("%-70s %8.3f GB/step (%.1f dispatches)" % (k[:70], v["hbm_bytes_per_step"] / 1e9, v["dispatches_per_step"]))
print("step total %.3f GB" % (doc["runs"]["%s/%s" % (wl, io)]["step_total_hbm_bytes"] / 1e9))
PY
