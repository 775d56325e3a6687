#!/bin/bash
mkdir -p gpurun_out
SMALTGPU_TIMING=1 timeout -k 10 500 python tools/bench_tool.py --reads 1000000 --paired > gpurun_out/r3_tool_paired.json 2> gpurun_out/r3_tool_paired.err; grep -v "^\[smalt," gpurun_out/r3_tool_paired.err | tail -8 | cut -c1-500; grep "smalt_gpu, g.fq" gpurun_out/r3_tool_paired.err | tail -12 | cut -c1-300; cut -c1-1200 gpurun_out/r3_tool_paired.json
timeout -k 10 600 python tools/bench_tool.py --reads 10000000 --native-repeat 3 --native-gap 10 > gpurun_out/r3_tool_10m_gap.json 2> gpurun_out/r3_tool_10m_gap.err; grep "smaltgpu-map:" gpurun_out/r3_tool_10m_gap.err | cut -c1-300
