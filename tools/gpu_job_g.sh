#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_large.py tests/test_gpu_fullsize_long.py tests/test_gpu_sw.py -x -q > gpurun_out/r3_strip_tests.log 2>&1; tail -3 gpurun_out/r3_strip_tests.log | cut -c1-600
bash tools/gpu_job_f.sh
