#!/bin/bash
# round-end evidence: the whole GPU suite, then the three bench lines with their CPU baselines
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_final_tests.log 2>&1; tail -3 gpurun_out/r3_final_tests.log | cut -c1-600
