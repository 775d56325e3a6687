#!/bin/bash
# rocprofv3 kernel statistics of smaltgpu-map on the bench inputs with extra options (e.g. -x or -p): tools/profile_native.sh <reads> <options...>
# -> gpurun_out/prof_native/nat_kernel_stats.csv and the program's own stage times
set -e
ROOT=$(pwd)
N=${1:-200000}; shift
T=$(mktemp -d /tmp/native.XXXX)
python - "$N" "$T" <<'PY'
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from smalt_amd import gpuindex, indexfile
n, tmp = int(sys.argv[1]), sys.argv[2]
dev = torch.device("cuda", 0)
nchr, chrlen, k, s, rlen = 24, 125000000, 13, 6, 150
sop = np.arange(nchr + 1, dtype=np.int64) * chrlen
ref = gpuindex.make_reference_gpu(nchr, chrlen, 20261004, dev)
packed = gpuindex.pack_reference(ref)
idx, pos = gpuindex.build_perfect_index(ref, sop, k, s)
reads, _ = gpuindex.make_reads_gpu(ref, sop, n, rlen, 777)
rd = reads.cpu().numpy().reshape(n, rlen)
prefix = os.path.join(tmp, "hs")
tot = int(sop[-1])
indexfile.write_sma(prefix, ["chr%d" % (i + 1) for i in range(nchr)], sop, packed.cpu().numpy())
indexfile.write_smi_perfect(prefix, k, s, idx.cpu().numpy(), pos.cpu().numpy(), (tot + s - 1) // s - 1)
q = b"I" * rlen
with open(os.path.join(tmp, "r.fq"), "wb") as f:
    for i in range(n):
        f.write(b"@r%d\n" % i + rd[i].tobytes() + b"\n+\n" + q + b"\n")
PY
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
SMALTGPU_MAP_VERBOSE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_native -o nat -- $ROOT/smalt_amd/smaltgpu-map -r -1 -f cigar -n 16 "$@" -o $T/out.cig $T/hs $T/r.fq 2>&1 | grep smaltgpu-map
rm -rf $T
