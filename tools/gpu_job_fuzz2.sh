#!/bin/bash
mkdir -p gpurun_out
for sd in 31 32; do
  FUZZ_BOUND=1 timeout -k 10 400 python tools/fuzz_pairs.py 40 3000 $sd > gpurun_out/r3_fzb_p$sd.log 2>&1; echo "bound pairs seed $sd: $(grep -c ' ok:' gpurun_out/r3_fzb_p$sd.log) ok, $(grep -c 'DIFFERS\|FAILED' gpurun_out/r3_fzb_p$sd.log) bad"
  FUZZ_BOUND=1 timeout -k 10 400 python tools/fuzz_single.py 40 3000 $sd > gpurun_out/r3_fzb_s$sd.log 2>&1; echo "bound single seed $sd: $(grep -c ' ok:' gpurun_out/r3_fzb_s$sd.log) ok, $(grep -c 'DIFFERS\|FAILED' gpurun_out/r3_fzb_s$sd.log) bad"
done
grep -h -A6 "DIFFERS\|FAILED" gpurun_out/r3_fzb_*.log | head -40 | cut -c1-300
