#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_large.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r3_f64sort_tests.log 2>&1; tail -3 gpurun_out/r3_f64sort_tests.log | cut -c1-600
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-buffers > gpurun_out/r3_f64sort_bench.json 2> gpurun_out/r3_f64sort_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r3_f64sort_bench.json")); print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
PY
