#!/bin/bash
mkdir -p gpurun_out
for g in 4096 5120 8192; do
  SMALTGPU_STRIP_GRID=$g timeout -k 10 200 python bench.py --long --reads 2000 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r3_long_g$g.json 2> gpurun_out/r3_long_g$g.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3_long_g$g.json")); print("grid $g", round(d["value"],1), d["roofline"]["gcups"], {k:round(v) for k,v in d["kernel_ms_per_step"].items() if v>1})
PY
done
