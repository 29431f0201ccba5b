#!/usr/bin/env python3
"""Print the table of DESIGN.md section 6 from the committed bench lines (profiles/r4_bench_*.json[l])."""
import json
def load(f): return [json.loads(l) for l in open(f) if l.startswith('{')]
ns = load('profiles/r4_bench_ns.json')[0]; mem = load('profiles/r4_bench_ns_members.jsonl'); cfg = load('profiles/r4_bench_configs.jsonl'); dn = load('profiles/r4_bench_dense_bf16.jsonl')
ns_no, ns_m2, ns_m4 = mem
c1, c1no, c2b, c2m4, c2no, c2f, c3, c3no, c4, c4g, c4k, nsk, k17 = cfg
c2d, c2dm4, nsd, c1d, c3d = dn
def fmt(v): return "–" if v is None else ("%.0f" % v if v >= 1000 else "%.2f" % v)
def cells(d, frac=False):
    r = d['roofline']; k = r['kernels']; tf = r['whole_step_tflops']
    member = d['config']['workload'].split('[gather-sum passes: ')[-1].split(';')[0].split(']')[0] if '[gather-sum passes: ' in d['config']['workload'] else ''
    member = member.replace('two-limb f16 dense GEMM of radius ', 'split r').replace(', fp32 accuracy', '').replace('exact fp32 gather', 'exact')
    return "%s (%s) | %s%s | %.2f / %.2f, %.2f | %s | %s" % (fmt(d['ms_per_step']), fmt(r.get('steady_state_ms')), ("%.1f" % tf), (" (%d %%)" % round(100 * tf / 157.3) if frac else ""),
        k['gather_sum_fwd']['avg_ms'], k['gather_sum_dx']['avg_ms'], k['gather_dot']['avg_ms'], member, "ok" if (d.get('parity_gate') or {}).get('ok') else "–")
rows = [
 ("`ns` N=128 256→256 56×56 G=4 k=9, μ~U(−3,3) — **the headline**", ns, True, "37.02"),
 ("`ns --no-split` (exact gather for every call)", ns_no, True, "37.02"),
 ("`ns --mu-range 2`", ns_m2, True, "—"),
 ("`ns --mu-range 4` (offsets up to the layer's clip 3.99)", ns_m4, True, "—"),
 ("`c1` AlexNet conv2 N=64 96→256 27×27 G=4", c1, True, "2.28"),
 ("`c1 --no-split`", c1no, True, "2.28"),
 ("`c2 --io bf16` N=128 256→256 56×56 **G=6**, bf16 activations (fp32-accurate arithmetic)", c2b, True, "52.33"),
 ("`c2 --io bf16 --mu-range 4`", c2m4, True, "—"),
 ("`c2 --io bf16 --no-split`", c2no, True, "52.33"),
 ("`c2` (fp32 activations)", c2f, True, "53.47"),
 ("`c2 --io bf16 --dense` (opt-in bf16-PRODUCT dense forms, bf16 bar; |μ| ≤ 3)", c2d, False, "12.48"),
 ("`c2 --io bf16 --dense --mu-range 4` (9×9 members)", c2dm4, False, "—"),
 ("`ns --io bf16 --dense`", nsd, False, "12.35"),
 ("`c1 --io bf16 --dense`", c1d, False, "1.04"),
 ("`c3 --io bf16 --dense`", c3d, False, "14.2"),
 ("`c3` N=128 512→512 28×28 G=4", c3, True, "40.30"),
 ("`c3 --no-split`", c3no, True, "40.30"),
 ("`c4` N=16 256→256 512×512, 9 live units of 10, k=65, μ within ±17 (+ parameter gradients of 8 channels gated)", c4, True, "1148"),
 ("`c4 --offsets grid` the same layer, units on the 3×3 grid ± 1 px", c4g, True, "1072"),
 ("`c4k33` the same maps, k=33, μ within ±15", c4k, True, "1105"),
 ("`--shape 128,256,256,56,56,4,17,7` NS shape, k=17, μ within ±7 (bucket 8)", k17, True, "39.80"),
 ("`nsk65` NS shape under `max_kernel_size=65`, μ within ±3", nsk, True, "37.34"),
]
print("| workload | ms/step (steady state) | whole-step TF algorithmic (of the fp32 roof) | gather-sum fwd / ∂x, gather-dot (ms per pass) | gather-sum member | gate | round 3 ms |")
print("|---|---|---|---|---|---|---|")
for label, d, frac, r3 in rows:
    print("| %s | %s | %s |" % (label, cells(d, frac), r3))
print()
print("lib", ns['lib'], "| layer", ns['layer']['ms_per_step'], "| ns400", load('profiles/r4_bench_ns_400.json')[0]['ms_per_step'],
      "| dist1", load('profiles/r4_bench_dist1_rccl.json')[0]['ms_per_step'], "| gloo2", load('profiles/r4_bench_gloo2_rehearsal.json')[0]['ms_per_step'],
      "| cpu", ns['cpu_baseline']['value'], ns['cpu_baseline']['cores'], "| roofline", ns['roofline']['kernel'], ns['roofline']['frac'], "traffic", ns['roofline']['traffic'])
