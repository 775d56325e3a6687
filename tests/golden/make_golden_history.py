#!/usr/bin/env python3
"""Fixture for the serial-order mode (hit-list capacity carried from read to read): inputs and the dump of the UNMODIFIED
reference run serially over them (oracle/_ref/refdump keeps one RMap, i.e. one hit list, for the whole file like
`smalt map -n 0` does -- rmap.c:1123).

The reference holds 430 copies of a 170-base unit.  With -x (all seeds take part, not the rarest ones up to 16 384 hits:
hashhit.c:769-891 keeps ordinary runs below the boundary) a 100-base read from the unit collects ~19 k hits per strand on that
sequence, more than the 16 384 its own length gives the list (hashhit.c:1262-1288), so it takes the allocation-boundary protocol
(hashhit.c:1497, :1730-1741) -- unless a longer read came before it: a 260-base read grows the list to 49 152 entries, and the
same 100-base reads behind it gather all their hits.  The file has them in that order: unit reads, one long read, the same unit
reads again, then ordinary reads of mixed lengths.  Runs only in the build container (needs `make -C oracle ref`)."""
import gzip
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
TAG, K, S, OPTS = "g_k13s2_hist", 13, 2, ["-x"]


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    rng = np.random.default_rng(20261004)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rnd = lambda n: acgt[rng.integers(0, 4, size=n)].tobytes()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    unit = rnd(170)
    chr1 = b"".join(rnd(int(rng.integers(150, 260))) + unit for _ in range(430)) + rnd(300)
    chr2 = rnd(90_000)
    unit_reads = []
    for i in range(6):
        r = bytearray(unit[10 + 5 * i:110 + 5 * i])
        if i % 2:
            p = 30 + 7 * i
            r[p] = ord("A") if r[p] != ord("A") else ord("G")
        unit_reads.append(bytes(r) if i % 3 else bytes(r)[::-1].translate(comp))
    p0 = 40_000
    long_read = chr2[p0:p0 + 260]
    mixed = []
    for i in range(24):
        n = int(rng.integers(30, 200))
        p = int(rng.integers(0, len(chr2) - n))
        r = bytearray(chr2[p:p + n])
        for j in range(len(r)):
            if rng.random() < 0.02:
                r[j] = b"ACGT"[int(rng.integers(0, 4))]
        mixed.append(bytes(r) if i % 2 else bytes(r)[::-1].translate(comp))
    reads = unit_reads + [long_read] + unit_reads + mixed[:12] + unit_reads[:2] + mixed[12:]
    with tempfile.TemporaryDirectory() as tmp:
        fa, fq, pre = os.path.join(tmp, TAG + ".fa"), os.path.join(tmp, TAG + ".fq"), os.path.join(tmp, TAG)
        with open(fa, "wb") as f:
            for i, c in enumerate((chr1, chr2)):
                f.write(b">chr%d\n" % (i + 1))
                for o in range(0, len(c), 70):
                    f.write(c[o:o + 70] + b"\n")
        with open(fq, "wb") as f:
            for i, r in enumerate(reads):
                f.write(b"@h%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
        subprocess.run([os.path.join(REF, "smalt"), "index", "-k", str(K), "-s", str(S), pre, fa], check=True, capture_output=True)
        dump = subprocess.run([os.path.join(REF, "refdump")] + OPTS + [pre, fq], check=True, capture_output=True).stdout
        # the same reads, every one as the first read of a run: what a mapper without the serial-order mode reproduces
        fresh = []
        for i, r in enumerate(reads):
            one = os.path.join(tmp, "one.fq")
            with open(one, "wb") as f:
                f.write(b"@h%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
            d = subprocess.run([os.path.join(REF, "refdump")] + OPTS + [pre, one], check=True, capture_output=True).stdout
            fresh.append(d.replace(b"READ 0 ", b"READ %d " % i, 1))
        fresh = b"".join(fresh)
        for src, dst in ((fa, TAG + ".fa.gz"), (fq, TAG + ".fq.gz")):
            with gzip.GzipFile(os.path.join(HERE, dst), "wb", mtime=0) as g:
                g.write(open(src, "rb").read())
        for data, dst in ((dump, TAG + ".refdump.txt.gz"), (fresh, TAG + ".refdump_fresh.txt.gz")):
            with gzip.GzipFile(os.path.join(HERE, dst), "wb", mtime=0) as g:
                g.write(data)
    nd = sum(1 for a, b in zip(dump.split(b"\n"), fresh.split(b"\n")) if a != b)
    m = dict(tag=TAG, k=K, s=S, opts=" ".join(OPTS), nreads=len(reads), dump_lines=dump.count(b"\n"), fresh_lines=fresh.count(b"\n"), lines_that_differ=nd)
    json.dump(m, open(os.path.join(HERE, "manifest_history.json"), "w"), indent=1)
    print(m)


if __name__ == "__main__":
    sys.exit(main())
