#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/r3_final_bench.json 2> gpurun_out/r3_final_bench.err; cut -c1-400 gpurun_out/r3_final_bench.json
timeout -k 10 500 python bench.py --paired --steps 2 --warmup 1 > gpurun_out/r3_final_paired.json 2> gpurun_out/r3_final_paired.err; cut -c1-300 gpurun_out/r3_final_paired.json
timeout -k 10 500 python bench.py --long --steps 1 --warmup 1 > gpurun_out/r3_final_long.json 2> gpurun_out/r3_final_long.err; cut -c1-300 gpurun_out/r3_final_long.json
