#!/bin/bash
mkdir -p gpurun_out
for sd in 4 5 6 7; do
  timeout -k 10 400 python tools/fuzz_pairs.py 40 4000 $sd > gpurun_out/r3_fz_p$sd.log 2>&1; echo "pairs seed $sd: $(grep -c ' ok:' gpurun_out/r3_fz_p$sd.log) ok, $(grep -c 'DIFFERS\|FAILED' gpurun_out/r3_fz_p$sd.log) bad"
  timeout -k 10 400 python tools/fuzz_single.py 40 4000 $sd > gpurun_out/r3_fz_s$sd.log 2>&1; echo "single seed $sd: $(grep -c ' ok:' gpurun_out/r3_fz_s$sd.log) ok, $(grep -c 'DIFFERS\|FAILED' gpurun_out/r3_fz_s$sd.log) bad"
done
grep -h -A6 "DIFFERS\|FAILED" gpurun_out/r3_fz_*.log | head -40 | cut -c1-300
