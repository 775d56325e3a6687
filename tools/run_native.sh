#!/bin/bash
# Diagnostic: the bench reference + reads on disk once, then smaltgpu-map under several settings (stage times on stderr).
# usage: tools/run_native.sh <reads> ; settings come from the NATIVE_ENVS variable ("A=1 B=2;C=3"), extra options from NATIVE_ARGS
# (run-to-run spread on one box: 410-560 k reads/s for the same binary, mostly in the first device allocations)
set -e
N=${1:-2000000}
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
IFS=';' read -ra SETS <<< "${NATIVE_ENVS:-X=1}"
for e in "${SETS[@]}"; do
  echo "== $e"
  env $e SMALTGPU_MAP_VERBOSE=1 ./smalt_amd/smaltgpu-map -r -1 -f cigar -n 16 $NATIVE_ARGS -o $T/out.cig $T/hs $T/r.fq 2>&1 | grep smaltgpu-map
  rm -f $T/out.cig
done
rm -rf $T
