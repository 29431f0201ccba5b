#!/usr/bin/env python3
"""Copy what tools/steps_artefacts.txt left under gpurun_out/ into profiles/r4_* (bench lines, rocprof kernel stats, PMC traffic,
SQ counters, parity margins, soak record) and print a summary table."""
import json, os, shutil
def last_json_line(f):
    for l in reversed(open(f).read().splitlines()):
        if l.startswith('{') and '"metric"' in l: return l
    raise SystemExit("no result line in " + f)
G = 'gpurun_out/'
def dump(dst, tags):
    with open(dst, 'w') as fh:
        for t in tags: fh.write(last_json_line(G + '%s.log' % t) + '\n')
dump('profiles/r4_bench_ns.json', ['fa_bench_ns'])
dump('profiles/r4_bench_ns_400.json', ['fa_bench_ns400'])
dump('profiles/r4_bench_ns_members.jsonl', ['fa_bench_ns_nosplit', 'fa_bench_ns_m2', 'fa_bench_ns_m4'])
CFG = ['fa_bench_c1', 'fa_bench_c1_nosplit', 'fa_bench_c2', 'fa_bench_c2_m4', 'fa_bench_c2_nosplit', 'fa_bench_c2f32', 'fa_bench_c3', 'fa_bench_c3_nosplit',
       'fa_bench_c4', 'fa_bench_c4grid', 'fa_bench_c4k33', 'fa_bench_nsk65', 'fa_bench_k17']
dump('profiles/r4_bench_configs.jsonl', CFG)
DN = ['fa_bench_c2dense', 'fa_bench_c2dense_m4', 'fa_bench_nsdense', 'fa_bench_c1dense', 'fa_bench_c3dense']
dump('profiles/r4_bench_dense_bf16.jsonl', DN)
dump('profiles/r4_bench_gloo2_rehearsal.json', ['fa_gloo2'])
dump('profiles/r4_bench_gloo2_c3_rehearsal.json', ['fa_gloo2_c3'])
dump('profiles/r4_bench_dist1_rccl.json', ['fa_dist1'])
for t, n in [('fa_ns', 'ns'), ('fa_c4', 'c4'), ('fa_c3', 'c3'), ('fa_c1', 'c1'), ('fa_c2d', 'c2_dense_bf16')]:
    shutil.copy(G + '%s_kernel_stats.csv' % t, 'profiles/r4_%s_rocprofv3_kernel_stats.csv' % n)
shutil.copy(G + 'fa_ns_pmc_traffic.json', 'profiles/r4_pmc_traffic.json')
if os.path.exists(G + 'parity_margins.jsonl'): shutil.copy(G + 'parity_margins.jsonl', 'profiles/r4_parity_margins.jsonl')
if os.path.exists(G + 'soak_memory.json'): shutil.copy(G + 'soak_memory.json', 'profiles/r4_soak_memory.json')
d = json.load(open(G + 'fa_ns_pmc_counters.json'))
out = {"_comment": "rocprofv3 --pmc (four passes, no trace domains) of `python3 bench.py --no-cpu-baseline --no-layer --no-check --steps 2 --warmup 1 "
       "--steady-seconds 0`, MI355X, mean per dispatch of the kernels that do the work of an NS step (final build of round 4). "
       "cycles = SQ_BUSY_CYCLES / 32; matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles.", "kernels": {}}
for k, c in d.items():
    if c.get('SQ_INSTS_MFMA', 0) < 1e6: continue
    cyc = c['SQ_BUSY_CYCLES'] / 32
    out["kernels"][k] = {"counters": c, "derived": {
        "cycles_per_dispatch": cyc, "mfma_pipe_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc,
        "valu_non_mfma_per_mfma": (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"],
        "lds_cycles_per_lds_instruction": c["SQ_LDS_IDX_ACTIVE"] / max(c["SQ_INSTS_LDS"], 1), "lds_pipe_busy_frac": c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc,
        "lds_bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1),
        "wait_inst_any_over_wave_cycles": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], "wait_inst_lds_over_wave_cycles": c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"],
        "valu_mfma_coexec_cycles": c.get("SQ_VALU_MFMA_COEXEC_CYCLES")}}
json.dump(out, open('profiles/r4_pmc_sq_counters_ns.json', 'w'), indent=1)
for t in ['fa_bench_ns', 'fa_bench_ns400', 'fa_bench_ns_nosplit', 'fa_bench_ns_m2', 'fa_bench_ns_m4'] + CFG + DN + ['fa_gloo2', 'fa_gloo2_c3', 'fa_dist1']:
    d = json.loads(last_json_line(G + '%s.log' % t)); r = d.get('roofline') or {}
    print(t, d['ms_per_step'], 'steady', r.get('steady_state_ms'), 'TF', r.get('whole_step_tflops'), {n: (v['avg_ms'], v['tflops']) for n, v in r.get('kernels', {}).items()},
          'parity', (d.get('parity_gate') or {}).get('ok'), 'layer', (d.get('layer') or {}).get('ms_per_step'), 'traffic', r.get('traffic'), 'exposed', (d.get('comm') or {}).get('exposed_ms'), d['lib'])
for k, v in out["kernels"].items(): print(k[:60], {a: round(b, 4) if isinstance(b, float) else b for a, b in v["derived"].items()})
t = json.load(open('profiles/r4_pmc_traffic.json')); r = t['runs']['ns/f32']; print(r['src_sha256_16'], r['step_total_hbm_bytes'] / 1e9)
