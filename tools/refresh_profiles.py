import csv, json, collections
rows=list(csv.DictReader(open('gpurun_out/prof_main/bench_kernel_stats.csv')))
smg=[r for r in rows if 'smg::' in r['Name']]
oth=[r for r in rows if r not in smg]
tot=sum(float(r['TotalDurationNs']) for r in rows)
out=["# rocprofv3 --kernel-trace --stats  (round 1, final state of the round)",
"# command: rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_main -o bench -- python3 bench.py --steps 1 --warmup 1 --reads 524288 --no-cpu-baseline",
"# 2 passes (warmup + timed) x 2 sub-batches of 262144 reads = 4 launches per kernel; durations in microseconds",
"# (names shortened; non-smg kernels are torch's reference/index/read generation in setup)",
"%-62s %6s %14s %14s %7s" % ("kernel","calls","total_us","avg_us","pct")]
for r in sorted(smg,key=lambda r:-float(r['TotalDurationNs'])):
    n=r['Name'].split('(')[0].replace('void ','')
    out.append("%-62s %6s %14.1f %14.1f %7.2f" % (n[:62], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
out.append("%-62s %6d %14.1f %14s %7.2f" % ("(torch/rocprim setup kernels, aggregated)", sum(int(r['Calls']) for r in oth), sum(float(r['TotalDurationNs']) for r in oth)/1e3, "-", 100*sum(float(r['TotalDurationNs']) for r in oth)/tot))
open('profiles/r01_bench_kernel_stats.txt','w').write("\n".join(out)+"\n")
print("\n".join(out[5:9]))
open('profiles/r01_bench_kernel_stats.bench.json','w').write(open('gpurun_out/prof_main.log').read().strip().splitlines()[-1]+"\n")
def load(path, ctrs):
    d=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        n=r['Kernel_Name']
        if 'smg::' not in n or r['Counter_Name'] not in ctrs: continue
        n=n.split('(')[0].replace('void ','')
        d[n][r['Counter_Name']]+=float(r['Counter_Value'])
    return d
f=load('gpurun_out/pmc_f/f_counter_collection.csv',{'FETCH_SIZE'})
w=load('gpurun_out/pmc_w/w_counter_collection.csv',{'WRITE_SIZE'})
names=sorted(set(f)|set(w), key=lambda n:-(f.get(n,{}).get('FETCH_SIZE',0)))
out=["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only), round 1 final state",
"# command: rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --reads 131072 --sub-batch 131072 --no-cpu-baseline",
"# one launch per kernel (131072 reads).  Counter unit: KB.  FETCH_SIZE is NOT doubled here (the x2 gfx950 correction of",
"# MI355X_MICROARCH.md applies to wide coalesced streams; these kernels read 4-8 byte words at random offsets).",
"%-40s %8s %16s %16s" % ("kernel","launches","FETCH_SIZE_MB","WRITE_SIZE_MB")]
for n in names:
    out.append("%-40s %8d %16.1f %16.1f" % (n[:40], 1, f.get(n,{}).get('FETCH_SIZE',0)/1024, w.get(n,{}).get('WRITE_SIZE',0)/1024))
open('profiles/r01_pmc_hbm_traffic.txt','w').write("\n".join(out)+"\n")
def g(d,key,c): return sum(v[c] for n,v in d.items() if key in n)*1024
pj={"round":1,"reads_per_launch":131072,"unit":"bytes per launch",
 "source":"profiles/r01_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE, separate passes; FETCH_SIZE not doubled: random 4-8 byte reads)",
 "kernels":{"sw_full":{"fetch":g(f,'k_sw_full16<8, 19, 248>','FETCH_SIZE'),"write":g(w,'k_sw_full16<8, 19, 248>','WRITE_SIZE')},
            "cands":{"fetch":g(f,'k_cands','FETCH_SIZE'),"write":g(w,'k_cands','WRITE_SIZE')},
            "seed":{"fetch":g(f,'k_seed','FETCH_SIZE'),"write":g(w,'k_seed','WRITE_SIZE')},
            "align":{"fetch":g(f,'k_align','FETCH_SIZE'),"write":g(w,'k_align','WRITE_SIZE')}}}
json.dump(pj,open('profiles/pmc_traffic.json','w'),indent=1)
C=['SQ_INSTS_VALU','SQ_ACTIVE_INST_VALU','SQ_WAVE_CYCLES','SQ_BUSY_CYCLES','SQ_WAIT_INST_ANY','SQ_WAIT_ANY','SQ_INSTS_LDS','SQ_LDS_BANK_CONFLICT']
sq=load('gpurun_out/pmc_s/s_counter_collection.csv',set(C))
out=["# rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT  (own pass, --kernel-trace only), round 1 final state",
"# command: rocprofv3 --pmc ... --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --reads 131072 --sub-batch 131072 --no-cpu-baseline",
"# one launch per kernel (131072 reads).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md);",
"# valu_act = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of a resident wave's time in which it issues VALU), wait_inst = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES",
"%-38s %12s %12s %12s %12s %12s %12s %9s %9s" % ("kernel","INSTS_VALU","ACTIVE_VALU","WAVE_CYCLES","INSTS_LDS","LDS_BANKCONF","WAIT_ANY","valu_act","wait_inst")]
for n in sorted(sq, key=lambda n:-sq[n]['SQ_INSTS_VALU']):
    v=sq[n]; wc=max(v['SQ_WAVE_CYCLES'],1)
    out.append("%-38s %12.4g %12.4g %12.4g %12.4g %12.4g %12.4g %9.3f %9.3f" % (n[:38], v['SQ_INSTS_VALU'], v['SQ_ACTIVE_INST_VALU'], v['SQ_WAVE_CYCLES'], v['SQ_INSTS_LDS'], v['SQ_LDS_BANK_CONFLICT'], v['SQ_WAIT_ANY'], v['SQ_ACTIVE_INST_VALU']/wc, v['SQ_WAIT_INST_ANY']/wc))
open('profiles/r01_pmc_sq_valu_lds.txt','w').write("\n".join(out)+"\n")
print("\n".join(out[4:8]))
