#!/bin/bash
# Profiles of one round on the GPU box (run through gpurun from the repository root): kernel trace + stats of the bench,
# PMC passes (HBM traffic, SQ) in runs of their own, the VALU issue-rate microbenchmark and the HBM counter calibration.
# Raw output goes to gpurun_out/; tools/refresh_profiles.py <round> turns it into the summaries under profiles/.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$ROOT/tools/valu_rate > $OUT/valu_rate.txt 2>&1
echo "[profile] valu_rate done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_main -o bench -- python3 $ROOT/bench.py --steps 1 --warmup 1 --reads 524288 --no-cpu-baseline --no-host-buffers > $OUT/prof_main.log 2> $OUT/prof_main.err
echo "[profile] kernel trace done"
B="python3 $ROOT/bench.py --steps 1 --warmup 0 --reads 131072 --sub-batch 131072 --no-cpu-baseline --no-host-buffers"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_f -o f -- $B > $OUT/pmc_f.log 2>&1
echo "[profile] pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_w -o w -- $B > $OUT/pmc_w.log 2>&1
echo "[profile] pmc write done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_s -o s -- $B > $OUT/pmc_s.log 2>&1
echo "[profile] pmc sq done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_f -o f -- $ROOT/tools/hbm_calib > $OUT/cal_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cal_w -o w -- $ROOT/tools/hbm_calib > $OUT/cal_w.log 2>&1
echo "[profile] calibration done"
find $OUT -name "*.csv" | head -30
