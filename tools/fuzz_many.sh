#!/bin/bash
# several seeds of both fuzzers (GPU box): fuzz_many.sh <first seed> <number of seeds> <cases> <reads or pairs per case>
mkdir -p gpurun_out
for ((sd = $1; sd < $1 + $2; sd++)); do
  timeout -k 10 500 python tools/fuzz_pairs.py $3 $4 $sd > gpurun_out/fz_p$sd.log 2>&1; echo "pairs seed $sd: $(grep -c ' ok:' gpurun_out/fz_p$sd.log) ok, $(grep -c 'DIFFERS\|FAILED' gpurun_out/fz_p$sd.log) bad"
  timeout -k 10 500 python tools/fuzz_single.py $3 $4 $sd > gpurun_out/fz_s$sd.log 2>&1; echo "single seed $sd: $(grep -c ' ok:' gpurun_out/fz_s$sd.log) ok, $(grep -c 'DIFFERS\|FAILED' gpurun_out/fz_s$sd.log) bad"
done
grep -h -A6 "DIFFERS\|FAILED" gpurun_out/fz_[ps]*.log | head -60 | cut -c1-300
