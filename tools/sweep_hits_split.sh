#!/bin/bash
run() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-buffers > gpurun_out/r3_sw_$name.json 2> gpurun_out/r3_sw_$name.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/r3_sw_$name.json')); k=d['kernel_ms_per_step']; print('$name', round(d['value']), 'hits %.1f cands %.1f sum %.1f' % (k['hits'], k['cands'], k['hits']+k['cands']))"
}
run w768_h3 SMALTGPU_HITS_WINDOW=768 SMALTGPU_HITS_WAVES=3
run w1024_h3 SMALTGPU_HITS_WINDOW=1024 SMALTGPU_HITS_WAVES=3
run w1024_h4 SMALTGPU_HITS_WINDOW=1024 SMALTGPU_HITS_WAVES=4
run w1024_h2 SMALTGPU_HITS_WINDOW=1024 SMALTGPU_HITS_WAVES=2
run w768_c3_l512 SMALTGPU_HITS_WINDOW=768 SMALTGPU_HITS_WAVES=3 SMALTGPU_CANDS_WAVES=3 SMALTGPU_CANDS_LDS_HITS=512
run w768_c3_l720 SMALTGPU_HITS_WINDOW=768 SMALTGPU_HITS_WAVES=3 SMALTGPU_CANDS_WAVES=3 SMALTGPU_CANDS_LDS_HITS=720
run w768_c2_l512 SMALTGPU_HITS_WINDOW=768 SMALTGPU_HITS_WAVES=3 SMALTGPU_CANDS_LDS_HITS=512
run w768_c2_l1024 SMALTGPU_HITS_WINDOW=768 SMALTGPU_HITS_WAVES=3 SMALTGPU_CANDS_LDS_HITS=1024
