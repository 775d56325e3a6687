#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pair_report.py tests/test_gpu_pairs.py tests/test_gpu_dropin.py -x -q > gpurun_out/r3_pairs_tests.log 2>&1; tail -4 gpurun_out/r3_pairs_tests.log | cut -c1-1200
timeout -k 10 300 python bench.py --paired --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_paired_f.json 2> gpurun_out/r3_paired_f.err
python - <<PY
import json
d=json.load(open("gpurun_out/r3_paired_f.json")); print(d["value"], d["ms_per_step"], d.get("gpu_busy_fraction"), d.get("host_ms_per_step"), d.get("round_wall_ms_per_step"), d.get("two_streams"))
PY
