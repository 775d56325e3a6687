#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_dropin.py -x -q -k "serial" > gpurun_out/r3_serial.log 2>&1; tail -5 gpurun_out/r3_serial.log | cut -c1-1200
timeout -k 10 300 python bench.py --paired --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_paired_e.json 2> gpurun_out/r3_paired_e.err
python - <<PY
import json
d=json.load(open("gpurun_out/r3_paired_e.json")); print(d["value"], d["ms_per_step"], d.get("gpu_busy_fraction"), d.get("host_ms_per_step"), d.get("round_wall_ms_per_step"))
PY
