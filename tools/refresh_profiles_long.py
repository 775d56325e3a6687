"""Summaries under profiles/ from the raw rocprofv3 output of tools/profile_long.sh (gpurun_out/long_*): usage refresh_profiles_long.py <round>"""
import collections
import csv
import sys

RN = int(sys.argv[1]) if len(sys.argv) > 1 else 3
R = 'r%02d' % RN
CMD = "python3 bench.py --long --reads 1000 --steps 1 --warmup 0 --no-cpu-baseline"


def short(n):
    return n.split('(')[0].replace('void ', '')


rows = [r for r in csv.DictReader(open('gpurun_out/long_main/bench_kernel_stats.csv'))]
tot = sum(float(r['TotalDurationNs']) for r in rows)
smg = [r for r in rows if 'smg::' in r['Name']]
out = ["# rocprofv3 --kernel-trace --stats (round %d), BASELINE configs[4] shape: 1000 reads of 8 kbp vs 3 Gbp, k=20 s=13" % RN,
       "# command: rocprofv3 --kernel-trace --stats --output-format csv -- " + CMD,
       "# one launch per kernel (k_cands, k_align: two, the second pass over deferred reads); durations in microseconds",
       "%-50s %6s %14s %14s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
for r in sorted(smg, key=lambda r: -float(r['TotalDurationNs'])):
    out.append("%-50s %6s %14.1f %14.1f %7.2f" % (short(r['Name'])[:50], r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
oth = [r for r in rows if r not in smg]
out.append("%-50s %6d %14.1f %14s %7.2f" % ("(torch/rocprim setup kernels, aggregated)", sum(int(r['Calls']) for r in oth), sum(float(r['TotalDurationNs']) for r in oth) / 1e3, "-",
                                            100 * sum(float(r['TotalDurationNs']) for r in oth) / tot))
open('profiles/%s_long_kernel_stats.txt' % R, 'w').write("\n".join(out) + "\n")
open('profiles/%s_long_kernel_stats.bench.json' % R, 'w').write(open('gpurun_out/long_main.log').read().strip().splitlines()[-1] + "\n")
print("\n".join(out[3:9]))


def load(path, ctrs):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        n = r['Kernel_Name']
        if 'smg::' not in n or r['Counter_Name'] not in ctrs:
            continue
        d[short(n)][r['Counter_Name']] += float(r['Counter_Value'])
    return d


f = load('gpurun_out/long_f/f_counter_collection.csv', {'FETCH_SIZE'})
w = load('gpurun_out/long_w/w_counter_collection.csv', {'WRITE_SIZE'})
C = ['SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_INSTS_LDS', 'SQ_LDS_BANK_CONFLICT']
sq = load('gpurun_out/long_s/s_counter_collection.csv', set(C))
out = ["# rocprofv3 --pmc (separate passes: FETCH_SIZE; WRITE_SIZE; the SQ counters), --kernel-trace only, round %d" % RN,
       "# command: rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- " + CMD,
       "# FETCH_SIZE / WRITE_SIZE in MB as reported (KB counters; streams read at half their bytes, isolated small reads at 64 B per line: profiles/%s_hbm_calibration.txt)" % R,
       "# valu_act = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES (quad-cycle counters)",
       "%-34s %12s %12s %12s %12s %12s %12s %9s %9s" % ("kernel", "FETCH_MB", "WRITE_MB", "INSTS_VALU", "WAVE_CYCLES", "INSTS_LDS", "LDS_BANKCONF", "valu_act", "wait_any")]
for n in sorted(set(f) | set(w) | set(sq), key=lambda n: -sq.get(n, {}).get('SQ_INSTS_VALU', 0)):
    v = sq.get(n, collections.defaultdict(float))
    wc = max(v['SQ_WAVE_CYCLES'], 1)
    out.append("%-34s %12.1f %12.1f %12.4g %12.4g %12.4g %12.4g %9.3f %9.3f" % (n[:34], f.get(n, {}).get('FETCH_SIZE', 0) / 1024, w.get(n, {}).get('WRITE_SIZE', 0) / 1024, v['SQ_INSTS_VALU'],
                                                                          v['SQ_WAVE_CYCLES'], v['SQ_INSTS_LDS'], v['SQ_LDS_BANK_CONFLICT'], v['SQ_ACTIVE_INST_VALU'] / wc, v['SQ_WAIT_ANY'] / wc))
open('profiles/%s_long_pmc.txt' % R, 'w').write("\n".join(out) + "\n")
print("\n".join(out[4:10]))
