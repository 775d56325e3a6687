#!/usr/bin/env python3
"""BASELINE configs[4] shape (not the bench.py contract): PacBio-shape long reads (substitutions, insertions, deletions)
against a synthetic reference resident in HBM, k=20 s=13 by default (collision-type index, built by the library).
Times smaltgpu_map_batch on host buffers (PCIe included) over whole batches and, on a bounded sample, the unmodified
reference `smalt map -n T` with the index files the library saved.  Prints one JSON line."""
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
    ap.add_argument("-k", type=int, default=20)
    ap.add_argument("-s", type=int, default=13)
    ap.add_argument("--reads", type=int, default=2000)
    ap.add_argument("--read-len", type=int, default=8000)
    ap.add_argument("--batch", type=int, default=2000)
    ap.add_argument("--cpu-reads", type=int, default=48)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--sub", type=float, default=0.03)
    ap.add_argument("--ins", type=float, default=0.05)
    ap.add_argument("--del", dest="dele", type=float, default=0.04)
    a = ap.parse_args()
    import torch
    from smalt_amd import api, gpuindex, synth
    dev = torch.device("cuda", 0)
    chrlen = int(a.chr_mbp * 1e6)
    sop = [i * chrlen for i in range(a.nchr + 1)]
    names = ["chr%d" % (i + 1) for i in range(a.nchr)]
    ref = gpuindex.make_reference_gpu(a.nchr, chrlen, 20261004, dev)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    ascii_ref = torch.cat([lut[c.long()] for c in ref.split(1 << 28)])
    gix = api.Index.build_device(ascii_ref.data_ptr(), sop, names, a.k, a.s, 0)
    build_ms = gix.build_ms
    del ascii_ref
    # PacBio-shape reads: sources fetched from HBM, errors applied on the host (synth.make_long_reads on one pseudo-chromosome per read)
    rng = np.random.default_rng(4242)
    reads = []
    for i in range(a.reads):
        c = int(rng.integers(0, a.nchr)); p = int(rng.integers(0, chrlen - a.read_len - 8))
        src = ref[c * chrlen + p: c * chrlen + p + a.read_len + 4].cpu().numpy()
        r, _ = synth.make_long_reads([src], 1, a.read_len, seed=1000 + i, sub=a.sub, ins=a.ins, dele=a.dele)
        reads.append(synth.codes_to_ascii(r[0]))
    del ref
    torch.cuda.empty_cache()
    maxlen = max(len(r) for r in reads)
    par = gix.default_params()
    mp = api.Mapper(gix, min(a.batch, a.reads), maxlen)
    quals = [b"5" * len(r) for r in reads]
    nb = min(a.batch, a.reads)
    mp.map_batch(reads[:nb], quals[:nb], par, allow_read_errors=True)            # warm-up
    t = time.time()
    nmapped = nres = nerr = 0
    kms = {}
    for b0 in range(0, a.reads, nb):
        res, stats = mp.map_batch(reads[b0:b0 + nb], quals[b0:b0 + nb], par, allow_read_errors=True)
        nmapped += sum(1 for r in res if r); nres += sum(len(r) for r in res); nerr += sum(1 for s_ in stats if s_["err"])
        ms, _ = mp.timers()
        for kk, v in ms.items():
            kms[kk] = kms.get(kk, 0.0) + v
    wall = time.time() - t
    mp.close()
    out = {"what": "BASELINE configs[4] shape: PacBio-shape long reads, host buffers (PCIe included)", "reference_bases": sop[-1], "k": a.k, "s": a.s,
           "index": {"typ": int(gix.info().typ), "build_ms": build_ms}, "reads": a.reads, "read_len_source": a.read_len,
           "errors": {"sub": a.sub, "ins": a.ins, "del": a.dele}, "batch": nb,
           "gpu_reads_per_s": a.reads / wall, "gpu_bases_per_s": sum(len(r) for r in reads) / wall, "mapped_fraction": nmapped / a.reads,
           "alignments": nres, "reads_with_error": nerr, "kernel_ms": {kk: round(v, 1) for kk, v in kms.items()}}
    # the unmodified reference on a sample, same index files
    smalt = os.path.join(ROOT, "oracle", "_ref", "smalt")
    if a.cpu_reads > 0 and os.path.exists(smalt):
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            prefix = os.path.join(tmp, "ix")
            gix.save(prefix)

            def fq(path, n):
                with open(path, "wb") as f:
                    for i in range(n):
                        f.write(b"@r%d\n" % i + reads[i] + b"\n+\n" + quals[i] + b"\n")

            def run(n):
                p = os.path.join(tmp, "r%d.fq" % n)
                fq(p, n)
                t0 = time.time()
                r = subprocess.run([smalt, "map", "-n", str(a.threads), "-f", "cigar", "-o", os.path.join(tmp, "o.cig"), prefix, p], capture_output=True)
                if r.returncode:
                    raise SystemExit("smalt map failed: " + r.stderr.decode()[-800:])
                return time.time() - t0
            n1 = max(2, a.cpu_reads // 6)
            t1, t2 = run(n1), run(a.cpu_reads)
            rate = (a.cpu_reads - n1) / max(t2 - t1, 1e-6)
            out["cpu_reference"] = {"kind": "reference", "threads": a.threads, "reads_per_s": rate, "sample": "%d vs %d reads (slope: index load cancels), %.1f s + %.1f s" % (n1, a.cpu_reads, t1, t2)}
            out["speedup"] = out["gpu_reads_per_s"] / rate
    gix.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
