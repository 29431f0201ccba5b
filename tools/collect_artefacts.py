#!/usr/bin/env python3
"""Copy what tools/steps_artefacts.txt left under gpurun_out/ into profiles/r3_* (bench lines, rocprof kernel stats, PMC traffic,
C4 counters, parity margins) and print a summary table."""
import json, os, shutil
def last_json_line(f):
    for l in reversed(open(f).read().splitlines()):
        if l.startswith('{') and '"metric"' in l: return l
G = 'gpurun_out/'
open('profiles/r3_bench_ns.json','w').write(last_json_line(G+'fa_bench_ns.log')+'\n')
open('profiles/r3_bench_ns_400.json','w').write(last_json_line(G+'fa_bench_ns400.log')+'\n')
with open('profiles/r3_bench_configs.jsonl','w') as fh:
    for t in ['fa_bench_c1','fa_bench_c2','fa_bench_c2f32','fa_bench_c3','fa_bench_c4','fa_bench_c4grid','fa_bench_c4k33','fa_bench_nsk65','fa_bench_k17']:
        fh.write(last_json_line(G+'%s.log'%t)+'\n')
with open('profiles/r3_bench_dense_bf16.jsonl','w') as fh:
    for t in ['fa_bench_c2dense','fa_bench_nsdense','fa_bench_c2dense_nosigma']:
        fh.write(last_json_line(G+'%s.log'%t)+'\n')
open('profiles/r3_bench_gloo2_rehearsal.json','w').write(last_json_line(G+'fa_gloo2.log')+'\n')
open('profiles/r3_bench_dist1_rccl.json','w').write(last_json_line(G+'fa_dist1.log')+'\n')
for t,n in [('fa_ns','ns'),('fa_c4','c4'),('fa_c3','c3'),('fa_c1','c1'),('fa_c2d','c2_dense_bf16')]:
    shutil.copy(G+'%s_kernel_stats.csv'%t,'profiles/r3_%s_rocprofv3_kernel_stats.csv'%n)
shutil.copy(G+'fa_ns_pmc_traffic.json','profiles/r3_pmc_traffic.json')
if os.path.exists(G+'parity_margins.jsonl'): shutil.copy(G+'parity_margins.jsonl','profiles/r3_parity_margins.jsonl')
d=json.load(open(G+'fa_c4_pmc_counters.json'))
k=[x for x in d if 'gather_dot' in x and d[x].get('SQ_INSTS_MFMA',0)>1e6][0]
c=d[k]; cyc=c['SQ_BUSY_CYCLES']/32
need_pass=4*9*16*262144*65536/256.0
ndisp=round(need_pass/0.7565/c['SQ_INSTS_MFMA'])
out={"_comment":"rocprofv3 --pmc (three passes, no trace domains) of `python3 bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-layer`, MI355X, mean per dispatch of the bucket-18 window-pass kernel (a C4 parameter-gradient pass runs in batch slabs under the 12 GB workspace budget: %d dispatches). The shipped form: work list of half-sweeps, greedy conflict-aware placement, empty lanes broadcast, flush every 1024 products, one workgroup per work list walking its rounds, error tile read in column groups. Compare profiles/r3_pmc_sq_counters_c4_before_worklist.json." % ndisp,
 "kernel":k,"counters":c,
 "derived":{"cycles_per_dispatch":cyc,"mfma_pipe_busy_frac":c["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/cyc,
  "valu_non_mfma_per_mfma":(c["SQ_INSTS_VALU"]-c["SQ_INSTS_MFMA"])/c["SQ_INSTS_MFMA"],
  "issue_cycles_frac (4 per packed VALU + 8 per MFMA, per SIMD)":((c["SQ_INSTS_VALU"]-c["SQ_INSTS_MFMA"])*4+c["SQ_INSTS_MFMA"]*8)/1024/cyc,
  "lds_instructions_per_mfma":c["SQ_INSTS_LDS"]/c["SQ_INSTS_MFMA"],
  "lds_cycles_per_lds_instruction":c["SQ_LDS_IDX_ACTIVE"]/c["SQ_INSTS_LDS"],"lds_pipe_busy_frac":c["SQ_LDS_IDX_ACTIVE"]/256/cyc,
  "wait_inst_any_over_wave_cycles":c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"],"wait_inst_lds_over_wave_cycles":c["SQ_WAIT_INST_LDS"]/c["SQ_WAVE_CYCLES"],
  "waves_resident_per_simd (SQ_WAVE_CYCLES counts quad cycles)":4*c["SQ_WAVE_CYCLES"]/(cyc*1024),
  "lane_utilisation (MFMAs needed per pass if every lane had a unit / MFMAs issued per pass)":need_pass/(ndisp*c['SQ_INSTS_MFMA'])}}
json.dump(out,open('profiles/r3_pmc_sq_counters_c4.json','w'),indent=1)
for t in ['fa_bench_ns','fa_bench_ns400','fa_bench_c1','fa_bench_c2','fa_bench_c2dense','fa_bench_c2f32','fa_bench_c3','fa_bench_c4','fa_bench_c4grid','fa_bench_c4k33','fa_bench_nsk65','fa_bench_nsdense','fa_bench_c2dense_nosigma','fa_bench_k17','fa_gloo2','fa_dist1']:
    d=json.loads(last_json_line(G+'%s.log'%t)); r=d.get('roofline') or {}
    print(t, d['ms_per_step'], 'steady',r.get('steady_state_ms'), 'TF',r.get('whole_step_tflops'), {n:(v['avg_ms'],v['tflops']) for n,v in r.get('kernels',{}).items()}, 'parity',(d.get('parity_gate') or {}).get('ok'), 'layer',(d.get('layer') or {}).get('ms_per_step'), 'traffic', r.get('traffic'), d['lib'])
print(json.dumps(out['derived'],indent=1))
t=json.load(open('profiles/r3_pmc_traffic.json')); r=t['runs']['ns/f32']; print(r['src_sha256_16'], r['step_total_hbm_bytes']/1e9)
