#!/usr/bin/env python3
"""Index construction: smaltgpu_index_build_device on a synthetic reference resident in HBM against
the reference's own single-threaded `smalt index` (oracle/_ref/smalt) on a bounded sample written to disk.
Prints one JSON line.  Not part of the bench.py contract."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nchr", type=int, default=24)
    ap.add_argument("--chr-mbp", type=float, default=125.0)
    ap.add_argument("-k", type=int, default=13)
    ap.add_argument("-s", type=int, default=6)
    ap.add_argument("--cpu-mbp", type=float, default=100.0, help="size of the sample the CPU reference indexes")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch
    from smalt_amd import api, gpuindex
    dev = torch.device("cuda", 0)
    chrlen = int(a.chr_mbp * 1e6)
    ref = gpuindex.make_reference_gpu(a.nchr, chrlen, 20261004, dev)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    ascii_ref = lut[ref.long()] if ref.numel() < (1 << 31) else torch.cat([lut[c.long()] for c in ref.split(1 << 28)])
    sop = [i * chrlen for i in range(a.nchr + 1)]
    names = ["chr%d" % (i + 1) for i in range(a.nchr)]
    tot = sop[-1]
    times = []
    info = None
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t = time.time()
        ix = api.Index.build_device(ascii_ref.data_ptr(), sop, names, a.k, a.s, 0)
        wall = time.time() - t
        d = ix.info()
        info = dict(typ=int(d.typ), npos=int(d.npos), nwords=int(d.nwords))
        times.append((ix.build_ms, wall * 1e3))
        ix.close()
    dev_ms = min(t[0] for t in times)
    # the same image from the torch-op builder used by bench.py (setup plumbing), for scale
    torch.cuda.synchronize()
    t = time.time()
    try:
        idx, pos = gpuindex.build_perfect_index(ref, np.asarray(sop, dtype=np.int64), a.k, a.s)
        torch.cuda.synchronize()
        torch_ms = (time.time() - t) * 1e3
        del idx, pos
    except AssertionError:
        torch_ms = None
    # CPU reference on a bounded sample
    nsmp = int(a.cpu_mbp * 1e6)
    smp = ascii_ref[:nsmp].cpu().numpy().tobytes()
    smalt = os.path.join(ROOT, "oracle", "_ref", "smalt")
    cpu_s = None
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        fa = os.path.join(tmp, "s.fa")
        with open(fa, "wb") as f:
            f.write(b">chr1\n")
            for o in range(0, nsmp, 60):
                f.write(smp[o:o + 60] + b"\n")
        t = time.time()
        r = subprocess.run([smalt, "index", "-k", str(a.k), "-s", str(a.s), os.path.join(tmp, "s"), fa], capture_output=True)
        if r.returncode == 0:
            cpu_s = time.time() - t
    ntup = (tot + a.s - 1) // a.s
    # algorithmic HBM bytes of the construction (PERFECT, 32-bit keys): every base read twice (packing, k-mer words) and
    # 0.4 B/base of packed words written; per sampled k-mer the (key, serial) pair written once (8 B) and read + written by
    # each of the ceil((2k+1)/8) radix passes (16 B per pass), the sorted keys read once more for the run ends (4 B); per
    # key the idx word written, read and written by the scan (12 B)
    passes = (2 * a.k + 1 + 7) // 8
    bytes_alg = tot * 2.4 + ntup * (8 + passes * 16 + 4) + (4 ** a.k) * 12
    out = {"what": "index construction, reference resident in HBM", "bases": tot, "k": a.k, "s": a.s, "index": info,
           "gpu_build_ms": dev_ms, "gpu_build_wall_ms": min(t[1] for t in times), "gpu_bases_per_s": tot / (dev_ms / 1e3),
           "torch_op_builder_ms": torch_ms,
           "roofline": {"bound": "hbm", "achieved": bytes_alg / (dev_ms / 1e3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": bytes_alg / (dev_ms / 1e3) / 1e9 / 8000.0, "bytes_algorithmic": bytes_alg},
           "cpu_reference": {"kind": "reference", "what": "`smalt index` (one thread, FASTA parse included)", "sample_bases": nsmp,
                             "seconds": cpu_s, "bases_per_s": (nsmp / cpu_s) if cpu_s else None},
           "speedup_per_base": (tot / (dev_ms / 1e3)) / (nsmp / cpu_s) if cpu_s else None}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
