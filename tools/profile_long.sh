#!/bin/bash
# configs[4] shape under rocprofv3 (run through gpurun from the repository root): kernel stats and PMC passes of their own
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --long --reads 1000 --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/long_main -o bench -- $B > $OUT/long_main.log 2> $OUT/long_main.err
echo "[profile] long kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/long_f -o f -- $B > $OUT/long_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/long_w -o w -- $B > $OUT/long_w.log 2>&1
echo "[profile] long pmc hbm done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/long_s -o s -- $B > $OUT/long_s.log 2>&1
echo "[profile] long pmc sq done"
