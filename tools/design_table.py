#!/usr/bin/env python3
"""Print the table of DESIGN.md section 6 from the committed bench lines (profiles/r3_bench_*.json[l])."""
import json
def load(f): return [json.loads(l) for l in open(f) if l.startswith('{')]
ns = load('profiles/r3_bench_ns.json')[0]; cfg = load('profiles/r3_bench_configs.jsonl'); dn = load('profiles/r3_bench_dense_bf16.jsonl')
c1, c2b, c2f, c3, c4, c4g, c4k, nsk, k17 = cfg
def cells(d, frac=False, extra=""):
    r = d['roofline']; k = r['kernels']
    tf = r['whole_step_tflops']
    return "%s (%s) | %s%s | %.1f / %.1f, %.1f%s" % (fmt(d['ms_per_step']), fmt(r.get('steady_state_ms')), ("%.1f" % tf), (" (%d %%)" % round(100 * tf / 157.3) if frac else ""),
        k['gather_sum_fwd']['tflops'], k['gather_sum_dx']['tflops'], k['gather_dot']['tflops'], extra)
def fmt(v): return "–" if v is None else ("%.0f" % v if v >= 1000 else "%.2f" % v)
rows = [
 ("`ns` N=128 256→256 56×56 G=4 k=9", ns, True, "", "36.73"),
 ("`c1` AlexNet conv2 N=64 96→256 27×27 G=4", c1, True, "", "2.30"),
 ("`c2 --io bf16` N=128 256→256 56×56 **G=6**, bf16 activations", c2b, False, "", "51.40"),
 ("`c2 --io bf16 --dense --check 2 --check-params 16` (densified bf16 gather-sum AND parameter gradients, §5.5; parity-gated)", dn[0], False, " (algorithmic; executed: %.2f PF dense in the parameter-gradient GEMMs = %d %% of the bf16 roof)" % (dn[0]['roofline']['achieved'] / 1000, round(100 * dn[0]['roofline']['frac'])), "23.95"),
 ("`c2 --io bf16 --dense --no-dsigma` (side line: a step that does not ask for ∂σ, the layer's default)", dn[2], False, " (algorithmic)", "—"),
 ("`c2` (fp32 activations)", c2f, False, "", "53.22"),
 ("`c3` N=128 512→512 28×28 G=4", c3, True, "", "40.22"),
 ("`c4 --check 1 --check-params 8` N=16 256→256 512×512, 9 live units of 10, **k=65, μ within ±17** (parity-gated incl. parameter gradients)", c4, True, "", "1800 (31.3)"),
 ("`c4 --offsets grid --check 1 --check-params 8` the same layer with its nine units on the 3×3 grid the reference initialises them on (`DAUGridMean`) ± 1 px per channel pair", c4g, False, "", "1298 (52.7)"),
 ("`c4k33` the same maps, k=33, μ within ±15", c4k, False, "", "1713 (33.3)"),
 ("`--shape 128,256,256,56,56,4,17,7` NS shape, **k=17, μ within ±7** (bucket 8)", k17, False, "", "39.15"),
 ("`nsk65` NS shape under `max_kernel_size=65`, μ within ±3", nsk, False, "", "36.86"),
 ("`ns --io bf16 --dense --check 2 --check-params 16`", dn[1], False, " (algorithmic)", "23.8"),
]
print("| workload (`--workload`) | ms/step (steady state) | whole-step TF | gather-sum fwd / ∂x, gather-dot (TF) | round 2 (its box) |")
print("|---|---|---|---|---|")
for label, d, frac, extra, r2 in rows:
    print("| %s | %s | %s |" % (label, cells(d, frac, extra), r2))
print("lib", ns['lib'], "layer", ns['layer']['ms_per_step'], "ns400", load('profiles/r3_bench_ns_400.json')[0]['ms_per_step'],
      "dist1", load('profiles/r3_bench_dist1_rccl.json')[0]['ms_per_step'], "gloo2", load('profiles/r3_bench_gloo2_rehearsal.json')[0]['ms_per_step'], "cpu", ns['cpu_baseline']['value'])
