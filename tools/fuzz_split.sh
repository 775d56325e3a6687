#!/bin/bash
# split reads (-p) through both programs on the library: fuzz_split.sh <first seed> <number of seeds> <cases> <reads per case>   (GPU box)
mkdir -p gpurun_out
for ((sd = $1; sd < $1 + $2; sd++)); do
  FUZZ_SPLIT=1 timeout -k 10 500 python tools/fuzz_single.py $3 $4 $sd > gpurun_out/fz_sp$sd.log 2>&1; echo "split, smaltgpu-map, seed $sd: $(grep -c ' ok:' gpurun_out/fz_sp$sd.log) ok, $(grep -c 'DIFFERS\|FAILED' gpurun_out/fz_sp$sd.log) bad, $(grep -c 'rejects' gpurun_out/fz_sp$sd.log) rejected by the reference"
  FUZZ_SPLIT=1 FUZZ_BOUND=1 timeout -k 10 500 python tools/fuzz_single.py $3 $4 $sd > gpurun_out/fz_sb$sd.log 2>&1; echo "split, bound program, seed $sd: $(grep -c ' ok:' gpurun_out/fz_sb$sd.log) ok, $(grep -c 'DIFFERS\|FAILED' gpurun_out/fz_sb$sd.log) bad"
done
grep -h -A6 "DIFFERS\|FAILED" gpurun_out/fz_s[pb]*.log | head -60 | cut -c1-300
