#!/bin/bash
# Diagnostic: shader clock during the kernels of the bench = GRBM_GUI_ACTIVE cycles / kernel duration (own PMC pass, kernel trace only).
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_clk -o c -- python3 $ROOT/bench.py --steps 1 --warmup 0 --reads 131072 --sub-batch 131072 --no-cpu-baseline --no-host-buffers > $OUT/pmc_clk.log 2>&1
python3 - "$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
dur = {}
for r in csv.DictReader(open(out + "/pmc_clk/c_kernel_trace.csv")):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for r in csv.DictReader(open(out + "/pmc_clk/c_counter_collection.csv")):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "smg::" not in r["Kernel_Name"]:
        continue
    name, ns = dur.get(r["Dispatch_Id"], (r["Kernel_Name"], 0))
    if ns > 2_000_000:
        print("%-40s %10.2f ms  %14.0f cycles  %7.0f MHz" % (name.split("(")[0].replace("void ", "")[:40], ns / 1e6, float(r["Counter_Value"]), float(r["Counter_Value"]) / ns * 1e3))
PY
