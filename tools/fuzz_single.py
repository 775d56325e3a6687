#!/usr/bin/env python3
"""Randomised single-end configurations: `smalt map` (the unmodified reference, oracle/_ref) against `smaltgpu-map` (the library
alone) on the same read file -- reference shape, index word length and stride, read lengths, output format and search options
drawn per case.  Prints one line per case and the first differing lines; exit status 1 if any case differs.
usage: fuzz_single.py [ncases] [nreads] [seed]        (GPU box; needs make -C oracle ref)
FUZZ_BOUND=1 compares the bound program (oracle/_ref/smalt_gpu) instead of smaltgpu-map.
FUZZ_SPLIT=1: split reads (-p) on chimeric reads (tests/test_gpu_split.py), SSAHA lines and user scores (-S) among the draws."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_gpu_dropin as t  # noqa: E402
from smalt_amd import synth  # noqa: E402


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    prog = os.path.join(ROOT, "smalt_amd", "smaltgpu-map")
    bad = 0
    for case in range(ncases):
        rng = np.random.default_rng(seed0 * 1000 + case)
        k, s = [(13, 6), (11, 3), (13, 2), (12, 4), (14, 7), (9, 6), (16, 8), (20, 13)][int(rng.integers(0, 8))]
        concat = rng.random() < 0.25
        nchr, chrlen = (int(rng.integers(520, 900)), int(rng.integers(1000, 3000))) if concat else (int(rng.integers(1, 9)), int(rng.integers(60_000, 400_000)))
        rlen = int(rng.choice([36, 50, 75, 100, 150, 250, 400]))
        opts = ["-f", str(rng.choice(["cigar", "sam", "samsoft", "sam:nohead,x", "sam:clip"]))]
        if rng.random() < 0.35:
            opts += ["-x"] + (["-c", str(round(float(rng.uniform(0.2, 0.8)), 2)) if rng.random() < 0.7 else str(int(rng.integers(k + s, rlen)))] if rng.random() < 0.6 else [])
        if rng.random() < 0.3:
            opts += ["-q", str(int(rng.integers(2, 11)))]
        if rng.random() < 0.3:
            opts += ["-m", str(int(rng.integers(k + s, max(k + s + 1, rlen // 2))))]
        if rng.random() < 0.3:
            opts += ["-y", str(round(float(rng.uniform(0.6, 0.98)), 2))]
        opts += ["-r", str(rng.choice(["-1", "3", "11"]))]
        split = bool(os.environ.get("FUZZ_SPLIT"))
        if split:
            opts = ["-p"] + opts
            if rng.random() < 0.3:
                opts[opts.index("-f") + 1] = "ssaha"
            if rng.random() < 0.3:
                opts += ["-S", str(rng.choice(["match=2,subst=-3,gapopen=-5,gapext=-3", "subst=-3,gapopen=-6", "match=1,subst=-1,gapopen=-3,gapext=-2", "gapopen=-5,gapext=-4"]))]
        if rng.random() < 0.35:
            opts += ["-d", str(rng.choice(["0", "3", "10", "-1"]))]
        with tempfile.TemporaryDirectory() as tmp:
            ch = synth.make_reference(nchr, chrlen, seed=seed0 * 7919 + case, repeat_frac=0.0 if concat else float(rng.choice([0.0, 0.1, 0.3, 0.5])), n_fam=3, cons_len=300,
                                      divergence=float(rng.choice([0.0, 0.02, 0.05])))
            fa, fq = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fq")
            synth.write_fasta(fa, ch)
            reads, _ = synth.make_reads(ch, nreads, min(rlen, chrlen - 10), seed=seed0 * 104729 + case, sub_rate=float(rng.choice([0.0, 0.02, 0.05])), indel_read_frac=float(rng.choice([0.0, 0.2, 0.5])))
            if split and min(rlen, chrlen - 10) >= 60:
                import test_gpu_split as tgs
                reads = [np.frombuffer(bytes(b).translate(bytes.maketrans(b"ACGTN", bytes([0, 1, 2, 3, 0]))), dtype=np.uint8) for _, b in tgs.chimeric_reads(ch, nreads, min(rlen, chrlen - 10), seed0 * 15485863 + case)]
            with open(fq, "wb") as f:
                for i, r in enumerate(reads):
                    b = bytearray(synth.codes_to_ascii(r))
                    u = rng.random()
                    if u < 0.03:
                        b = bytearray(synth.codes_to_ascii(rng.integers(0, 4, size=len(b), dtype=np.uint8)))
                    elif u < 0.08:
                        b = b[:int(rng.integers(5, len(b)))]
                    if rng.random() < 0.05 and len(b) > 4:
                        b[int(rng.integers(0, len(b)))] = ord("N")
                    q = bytes(33 + int(x) for x in rng.integers(2, 41, size=len(b)))
                    f.write(b"@r%d\n" % i + bytes(b) + b"\n+\n" + q + b"\n")
            pre = os.path.join(tmp, "idx")
            r0 = subprocess.run([t.SMALT, "index", "-k", str(k), "-s", str(s), pre, fa], capture_output=True)
            if r0.returncode:
                print("case %d: the reference rejects the index k=%d s=%d" % (case, k, s), flush=True)
                continue
            ref_out, gpu_out = os.path.join(tmp, "ref.out"), os.path.join(tmp, "gpu.out")
            r0 = subprocess.run([t.SMALT, "map"] + opts + ["-o", ref_out, pre, fq], capture_output=True)
            if r0.returncode:
                print("case %d: the reference rejects %s" % (case, " ".join(opts)), flush=True)
                continue
            if os.environ.get("FUZZ_BOUND"):      # the reference's own program with its mapping worker bound to the library (integration/)
                thr = ["-n", "3", "-O"] if "-1" in opts[opts.index("-r") + 1] else []       # worker threads only without random draws (they share one generator)
                r1 = subprocess.run([t.SMALT_GPU, "map"] + opts + thr + ["-o", gpu_out, pre, fq], capture_output=True, env=dict(os.environ, SMALTGPU_INDEX_PREFIX=pre))
            else:
                r1 = subprocess.run([prog] + opts + ["-B", str(int(rng.integers(100, 2500))), "-o", gpu_out, pre, fq], capture_output=True)
            if r1.returncode:
                print("case %d FAILED to run: k=%d s=%d nchr=%d rlen=%d %s: %s" % (case, k, s, nchr, rlen, " ".join(opts), r1.stderr.decode()[-300:]), flush=True)
                bad += 1
                continue
            a = [ln for ln in open(ref_out).read().split("\n") if not ln.startswith("@PG")]
            b = [ln for ln in open(gpu_out).read().split("\n") if not ln.startswith("@PG")]
            diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
            ok = len(a) == len(b) and not diff
            print("case %d %s: k=%d s=%d nchr=%d rlen=%d %s -> %d lines, %d differ" % (case, "ok" if ok else "DIFFERS", k, s, nchr, rlen, " ".join(opts), len(a), len(diff)), flush=True)
            if not ok:
                bad += 1
                for d in diff[:3]:
                    print("   line %d\n     ref: %s\n     gpu: %s" % (d[0], d[1][:300], d[2][:300]), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
