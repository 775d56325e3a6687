#!/usr/bin/env python3
"""Randomised paired configurations: `smalt map` (the unmodified reference, oracle/_ref) against `smaltgpu-map` (the library
alone) on the same two read files -- reference shape, read length, insert distribution, library type, output format and
search options drawn per case.  Prints one line per case and the first differing lines; exit status 1 if any case differs.
usage: fuzz_pairs.py [ncases] [npairs] [seed]        (GPU box; needs make -C oracle ref)
FUZZ_BOUND=1 compares the bound program (oracle/_ref/smalt_gpu) instead of smaltgpu-map."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_gpu_dropin as t  # noqa: E402


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    prog = os.path.join(ROOT, "smalt_amd", "smaltgpu-map")
    bad = 0
    only = int(os.environ["FUZZ_ONLY"]) if "FUZZ_ONLY" in os.environ else None     # FUZZ_ONLY=<case> FUZZ_KEEP=<dir>: write that case's inputs and stop
    for case in range(ncases):
        if only is not None and case != only:
            continue
        rng = np.random.default_rng(seed0 * 1000 + case)
        k, s = [(13, 6), (11, 3), (13, 2), (12, 4), (14, 7)][int(rng.integers(0, 5))]
        concat = rng.random() < 0.25
        nchr, chrlen = (int(rng.integers(520, 700)), 2000) if concat else (int(rng.integers(1, 6)), int(rng.integers(120_000, 400_000)))
        rlen = int(rng.choice([50, 75, 100, 150, 200]))
        mean = int(rng.integers(2 * rlen, 4 * rlen + 100))
        sd = int(rng.integers(10, 60))
        lib = str(rng.choice(["pe", "mp", "pp"]))
        opts = ["-f", str(rng.choice(["cigar", "sam", "samsoft"])), "-i", str(mean + 4 * sd + int(rng.integers(0, 200)))]
        if rng.random() < 0.4:
            opts += ["-j", str(int(rng.integers(0, mean // 2)))]
        if lib != "pe":
            opts += ["-l", lib]
        if rng.random() < 0.3:
            opts += ["-x"] + (["-c", str(round(float(rng.uniform(0.2, 0.7)), 2))] if rng.random() < 0.5 else [])
        if rng.random() < 0.3:
            opts += ["-q", str(int(rng.integers(2, 11)))]          # the reference accepts 0 .. 10
        if rng.random() < 0.3:
            opts += ["-m", str(int(rng.integers(20, 60)))]
        if rng.random() < 0.3:
            opts += ["-y", str(round(float(rng.uniform(0.7, 0.98)), 2))]
        opts += ["-r", str(rng.choice(["-1", "3", "11"]))]
        if rng.random() < 0.2:
            opts += ["-d", str(rng.choice(["0", "5", "-1"]))]
        with tempfile.TemporaryDirectory() as tmp:
            if only is not None:
                tmp = os.environ["FUZZ_KEEP"]
                os.makedirs(tmp, exist_ok=True)
            fa, fqs = t._pair_data(tmp, nchr, chrlen, npairs, rlen, seed=seed0 * 7919 + case, rep=0.0 if concat else float(rng.choice([0.1, 0.3, 0.5])), ins=(mean, sd))
            if os.environ.get("FUZZ_SPLIT"):          # split reads (-p): two mates in five get a stretch from somewhere else, either strand
                opts = ["-p"] + opts
                seqs = [x.split(b"\n", 1)[1].replace(b"\n", b"") for x in open(fa, "rb").read().split(b">")[1:]]
                comp4 = bytes.maketrans(b"ACGT", b"TGCA")
                for which, fq in enumerate(fqs):
                    recs = open(fq, "rb").read().split(b"\n")
                    for i in range(0, len(recs) - 3, 4):
                        b = recs[i + 1]
                        if (i // 4 + which) % 5 < 2 and len(b) > 70:
                            cut = int(rng.integers(25, len(b) - 25))
                            sq = seqs[int(rng.integers(0, len(seqs)))]
                            if len(sq) <= len(b):
                                continue
                            p = int(rng.integers(0, len(sq) - len(b)))
                            alien = sq[p:p + len(b) - cut]
                            if rng.random() < 0.5:
                                alien = alien.translate(comp4)[::-1]
                            recs[i + 1] = (b[:cut] + alien) if rng.random() < 0.5 else (alien + b[len(b) - cut:])
                    open(fq, "wb").write(b"\n".join(recs))
            if lib != "pe":               # the generator makes FR pairs: turn the mates as the library type expects
                comp = bytes.maketrans(b"ACGTacgtN", b"TGCAtgcaN")
                recs = open(fqs[1], "rb").read().split(b"\n")
                for i in range(0, len(recs) - 3, 4):
                    recs[i + 1] = recs[i + 1][::-1].translate(comp)
                    recs[i + 3] = recs[i + 3][::-1]
                open(fqs[1], "wb").write(b"\n".join(recs))
                if lib == "mp":           # RF: swap the files' roles
                    fqs = [fqs[1], fqs[0]]
            pre = os.path.join(tmp, "idx")
            subprocess.run([t.SMALT, "index", "-k", str(k), "-s", str(s), pre, fa], check=True, capture_output=True)
            if only is not None:
                open(os.path.join(tmp, "case.txt"), "w").write("%d %d %s\n%s\n" % (k, s, " ".join(fqs), " ".join(opts)))
                print("case %d kept in %s: %s" % (case, tmp, " ".join(opts)))
                return 0
            ref_out, gpu_out = os.path.join(tmp, "ref.out"), os.path.join(tmp, "gpu.out")
            r0 = subprocess.run([t.SMALT, "map"] + opts + ["-o", ref_out, pre] + fqs, capture_output=True)
            if r0.returncode:
                print("case %d: the reference rejects %s" % (case, " ".join(opts)), flush=True)
                continue
            if os.environ.get("FUZZ_BOUND"):      # the reference's own program with its mapping worker bound to the library (integration/)
                thr = ["-n", "3", "-O"] if "-1" in opts[opts.index("-r") + 1] else []       # worker threads only without random draws (they share one generator)
                r1 = subprocess.run([t.SMALT_GPU, "map"] + opts + thr + ["-o", gpu_out, pre] + fqs, capture_output=True, env=dict(os.environ, SMALTGPU_INDEX_PREFIX=pre))
            else:
                r1 = subprocess.run([prog] + opts + ["-B", str(int(rng.integers(100, 1500))), "-o", gpu_out, pre] + fqs, capture_output=True)
            if r1.returncode:
                print("case %d FAILED to run: k=%d s=%d nchr=%d rlen=%d %s: %s" % (case, k, s, nchr, rlen, " ".join(opts), r1.stderr.decode()[-300:]), flush=True)
                bad += 1
                continue
            a = [ln for ln in open(ref_out).read().split("\n") if not ln.startswith("@PG")]
            b = [ln for ln in open(gpu_out).read().split("\n") if not ln.startswith("@PG")]
            diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
            ok = len(a) == len(b) and not diff
            print("case %d %s: k=%d s=%d nchr=%d rlen=%d insert N(%d,%d) %s -> %d lines, %d differ" % (case, "ok" if ok else "DIFFERS", k, s, nchr, rlen, mean, sd, " ".join(opts), len(a), len(diff)), flush=True)
            if not ok:
                bad += 1
                for d in diff[:3]:
                    print("   line %d\n     ref: %s\n     gpu: %s" % (d[0], d[1][:300], d[2][:300]), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
