"""Summaries under profiles/ from the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/): usage refresh_profiles.py <round>"""
import csv, json, collections, sys, os
RN = int(sys.argv[1]) if len(sys.argv) > 1 else 1
R = 'r%02d' % RN
rows=list(csv.DictReader(open('gpurun_out/prof_main/bench_kernel_stats.csv')))
smg=[r for r in rows if 'smg::' in r['Name']]
oth=[r for r in rows if r not in smg]
tot=sum(float(r['TotalDurationNs']) for r in rows)
out=["# rocprofv3 --kernel-trace --stats  (round %d)" % RN,
"# command: rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_main -o bench -- python3 bench.py --steps 1 --warmup 1 --reads 524288 --no-cpu-baseline --no-host-buffers",
"# 2 passes (warmup + timed) x 2 sub-batches of 262144 reads = 4 launches per kernel; durations in microseconds",
"# (k_cands and k_align: 8 -- each is followed by its second pass over the reads the first one deferred, a launch that ends at",
"#  once when there are none: their time per sub-batch is total_us / 4)",
"# (names shortened; non-smg kernels are torch's reference/index/read generation in setup)",
"%-62s %6s %14s %14s %7s" % ("kernel","calls","total_us","avg_us","pct")]
for r in sorted(smg,key=lambda r:-float(r['TotalDurationNs'])):
    n=r['Name'].split('(')[0].replace('void ','')
    out.append("%-62s %6s %14.1f %14.1f %7.2f" % (n[:62], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
out.append("%-62s %6d %14.1f %14s %7.2f" % ("(torch/rocprim setup kernels, aggregated)", sum(int(r['Calls']) for r in oth), sum(float(r['TotalDurationNs']) for r in oth)/1e3, "-", 100*sum(float(r['TotalDurationNs']) for r in oth)/tot))
open('profiles/%s_bench_kernel_stats.txt' % R,'w').write("\n".join(out)+"\n")
print("\n".join(out[5:9]))
open('profiles/%s_bench_kernel_stats.bench.json' % R,'w').write(open('gpurun_out/prof_main.log').read().strip().splitlines()[-1]+"\n")
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
out=["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only), round %d" % RN,
"# command: rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --reads 131072 --sub-batch 131072 --no-cpu-baseline",
"# one launch per kernel (131072 reads).  Counter unit: KB, shown in MB.  FETCH_SIZE as reported.  Calibration on known byte counts",
"# (tools/hbm_calib.hip, profiles/%s_hbm_calibration.txt): a coalesced 16 B/lane stream reports 0.50 of its bytes (the guide's x2" % R,
"# correction); an isolated 8-byte probe reports 64 B and a random 32-byte list chunk 78 B -- one 64-B request per line touched,",
"# so for these kernels the counter is a count of lines touched x 64 B, 8x / 2.4x the bytes the algorithm asked for.",
"%-40s %8s %16s %16s" % ("kernel","launches","FETCH_SIZE_MB","WRITE_SIZE_MB")]
for n in names:
    out.append("%-40s %8d %16.1f %16.1f" % (n[:40], 1, f.get(n,{}).get('FETCH_SIZE',0)/1024, w.get(n,{}).get('WRITE_SIZE',0)/1024))
open('profiles/%s_pmc_hbm_traffic.txt' % R,'w').write("\n".join(out)+"\n")
def g(d,key,c): return sum(v[c] for n,v in d.items() if key in n)*1024
pj={"round":RN,"reads_per_launch":131072,"unit":"bytes per launch",
 "source":"profiles/%s_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE, separate passes; FETCH_SIZE as reported = 64 B per line touched for the random 4-8 byte reads of these kernels, see profiles/%s_hbm_calibration.txt)" % (R, R),
 "kernels":{"sw_full":{"fetch":g(f,'k_sw_full16<8, 19, 248>','FETCH_SIZE'),"write":g(w,'k_sw_full16<8, 19, 248>','WRITE_SIZE')},
            "cands":{"fetch":g(f,'k_cands','FETCH_SIZE'),"write":g(w,'k_cands','WRITE_SIZE')},
            "hits":{"fetch":g(f,'k_hits','FETCH_SIZE'),"write":g(w,'k_hits','WRITE_SIZE')},
            "seed":{"fetch":g(f,'k_seed','FETCH_SIZE'),"write":g(w,'k_seed','WRITE_SIZE')},
            "align":{"fetch":g(f,'k_align','FETCH_SIZE'),"write":g(w,'k_align','WRITE_SIZE')}}}
json.dump(pj,open('profiles/pmc_traffic.json','w'),indent=1)
C=['SQ_INSTS_VALU','SQ_ACTIVE_INST_VALU','SQ_WAVE_CYCLES','SQ_BUSY_CYCLES','SQ_WAIT_INST_ANY','SQ_WAIT_ANY','SQ_INSTS_LDS','SQ_LDS_BANK_CONFLICT']
sq=load('gpurun_out/pmc_s/s_counter_collection.csv',set(C))
out=["# rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT  (own pass, --kernel-trace only), round %d" % RN,
"# command: rocprofv3 --pmc ... --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --reads 131072 --sub-batch 131072 --no-cpu-baseline",
"# one launch per kernel (131072 reads).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md);",
"# valu_act = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of a resident wave's time in which it issues VALU), wait_inst = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES",
"%-38s %12s %12s %12s %12s %12s %12s %9s %9s" % ("kernel","INSTS_VALU","ACTIVE_VALU","WAVE_CYCLES","INSTS_LDS","LDS_BANKCONF","WAIT_ANY","valu_act","wait_inst")]
for n in sorted(sq, key=lambda n:-sq[n]['SQ_INSTS_VALU']):
    v=sq[n]; wc=max(v['SQ_WAVE_CYCLES'],1)
    out.append("%-38s %12.4g %12.4g %12.4g %12.4g %12.4g %12.4g %9.3f %9.3f" % (n[:38], v['SQ_INSTS_VALU'], v['SQ_ACTIVE_INST_VALU'], v['SQ_WAVE_CYCLES'], v['SQ_INSTS_LDS'], v['SQ_LDS_BANK_CONFLICT'], v['SQ_WAIT_ANY'], v['SQ_ACTIVE_INST_VALU']/wc, v['SQ_WAIT_INST_ANY']/wc))
open('profiles/%s_pmc_sq_valu_lds.txt' % R,'w').write("\n".join(out)+"\n")
print("\n".join(out[4:8]))

# HBM counter calibration (tools/hbm_calib.hip) and the VALU issue-rate microbenchmark (tools/valu_rate.hip)
if os.path.exists('gpurun_out/cal_f/f_counter_collection.csv'):
    PAY = float(1 << 28)
    def one(path, ctr):
        d = collections.defaultdict(float)
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] == ctr: d[r['Kernel_Name'].split('(')[0]] += float(r['Counter_Value']) * 1024
        return d
    cf, cw = one('gpurun_out/cal_f/f_counter_collection.csv', 'FETCH_SIZE'), one('gpurun_out/cal_w/w_counter_collection.csv', 'WRITE_SIZE')
    unit = {'k_stream_read16': 16, 'k_probe8': 8, 'k_list4': 32, 'k_stream_write16': 16, 'k_record48': 48}
    what = {'k_stream_read16': 'coalesced 16 B/lane streaming read (the guide\'s reference pattern)', 'k_probe8': 'independent random 8-byte reads of a 1 GiB table (k_seed: idx probes)',
            'k_list4': '8 consecutive 4-byte words at a random offset, one per lane (k_cands: position lists)', 'k_stream_write16': 'coalesced 16 B/lane streaming write',
            'k_record48': 'one 48-byte record per lane, consecutive (ranked-candidate pool)'}
    out = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- tools/hbm_calib (round %d): every kernel moves 256 MiB of payload exactly once" % RN,
           "# counter / payload: what the counter reports per byte the kernel asked for; per access: counter bytes per access of the pattern",
           "%-18s %-10s %14s %10s %12s  %s" % ("kernel", "counter", "reported_MB", "ratio", "B/access", "pattern")]
    for k in ('k_stream_read16', 'k_probe8', 'k_list4'):
        out.append("%-18s %-10s %14.1f %10.3f %12.1f  %s" % (k, 'FETCH_SIZE', cf[k] / 1e6, cf[k] / PAY, cf[k] / PAY * unit[k], what[k]))
    for k in ('k_stream_write16', 'k_record48'):
        out.append("%-18s %-10s %14.1f %10.3f %12.1f  %s" % (k, 'WRITE_SIZE', cw[k] / 1e6, cw[k] / PAY, cw[k] / PAY * unit[k], what[k]))
    out += ["# => FETCH_SIZE halves wide streams (x2 correction applies to them) and counts one 64-B request per line touched for isolated small reads;",
            "#    WRITE_SIZE is exact for streams and for 48-byte records."]
    open('profiles/%s_hbm_calibration.txt' % R, 'w').write("\n".join(out) + "\n")
    print("\n".join(out))
if os.path.exists('gpurun_out/valu_rate.txt'):
    open('profiles/%s_valu_rate.txt' % R, 'w').write("# raw output of tools/valu_rate (hipcc --offload-arch=gfx950 tools/valu_rate.hip), round %d: 8 independent chains per lane, 8 waves per SIMD\n" % RN + open('gpurun_out/valu_rate.txt').read())
