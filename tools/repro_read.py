#!/usr/bin/env python3
"""Diagnostic: map reads [lo, hi) of tools/bench_tool.py's synthetic read set (same reference and seeds) through api.Mapper in
batches of --batch reads and print every read that comes back with an error code -- to chase a per-read failure that the
bound program reported on a long run (`... when processing read No. N`)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4000000)
    ap.add_argument("--lo", type=int, default=1900000)
    ap.add_argument("--hi", type=int, default=1930000)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--cands-per-read", type=int, default=1024)
    ap.add_argument("--nchr", type=int, default=24)
    ap.add_argument("--chr-mbp", type=float, default=125.0)
    a = ap.parse_args()
    import torch
    from smalt_amd import api, gpuindex
    dev = torch.device("cuda", 0)
    k, s, rlen = 13, 6, 150
    chrlen = int(a.chr_mbp * 1e6)
    sop = np.arange(a.nchr + 1, dtype=np.int64) * chrlen
    ref = gpuindex.make_reference_gpu(a.nchr, chrlen, 20261004, dev)
    reads, _ = gpuindex.make_reads_gpu(ref, sop, a.reads, rlen, 777)
    rd = reads.cpu().numpy().reshape(a.reads, rlen)[a.lo:a.hi].copy()
    del reads
    ascii_ref = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[ref.long()]
    torch.cuda.synchronize()
    gix = api.Index.build_device(ascii_ref.data_ptr(), [int(x) for x in sop], ["chr%d" % (i + 1) for i in range(a.nchr)], k, s, 0)
    del ref, ascii_ref
    torch.cuda.empty_cache()
    mp = api.Mapper(gix, 16384, rlen, cands_per_read=a.cands_per_read)
    par = gix.default_params()
    bad = []
    for b0 in range(0, rd.shape[0], a.batch):
        blk = [bytes(x) for x in rd[b0:b0 + a.batch]]
        res, stats = mp.map_batch(blk, [b"I" * rlen] * len(blk), par, allow_read_errors=True)
        for i, st in enumerate(stats):
            if st["err"]:
                bad.append(dict(read=a.lo + b0 + i, err=st["err"], seq=blk[i].decode(), stats={k_: int(v) for k_, v in st.items()}))
    for x in bad:                           # each failing read once more, on its own
        res, stats = mp.map_batch([x["seq"].encode()], [b"I" * rlen], par, allow_read_errors=True)
        x["alone_err"] = stats[0]["err"]
        x["last_error"] = api.lib().smaltgpu_last_error().decode()
    print(json.dumps(dict(checked=[a.lo, a.hi], failing=bad)))
    mp.close()
    gix.close()


if __name__ == "__main__":
    main()
