#!/bin/bash
# GPU box job (round 3): two-rank test, long-read bench, paired bench with and without sub-blocks
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -x -q -k "two_ranks" > gpurun_out/r3_two_ranks.log 2>&1; tail -3 gpurun_out/r3_two_ranks.log | cut -c1-600
for n in 1 2; do
  SMALTGPU_PAIR_SUBBLOCKS=$n timeout -k 10 300 python bench.py --paired --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_paired_sub$n.json 2> gpurun_out/r3_paired_sub$n.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3_paired_sub$n.json")); print("sub$n", d["value"], d["ms_per_step"], d.get("gpu_busy_fraction"), d.get("two_streams"))
PY
done
timeout -k 10 900 python bench.py --long --steps 1 --warmup 1 > gpurun_out/r3_long.json 2> gpurun_out/r3_long.err; tail -3 gpurun_out/r3_long.err | cut -c1-400; cut -c1-1500 gpurun_out/r3_long.json
