#!/usr/bin/env python3
"""Generate the committed golden fixtures from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference and `make -C oracle ref`).  For each
small seeded configuration it writes the inputs (FASTA, FASTQ), the md5 of the index files
produced by the reference's `smalt index`, and the per-stage dump printed by
oracle/_ref/refdump (our driver around the reference's own functions; format in
oracle/DUMPFORMAT.md).  Fixtures are data only: inputs and expected outputs.
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from smalt_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")

CONFIGS = [
    # tag, nchr, chrlen, k, s, nreads, rlen, repeat_frac, refdump opts, flavour
    dict(tag="g_k13s6_hash", nchr=3, chrlen=120000, k=13, s=6, nreads=250, rlen=150, rep=0.15, opts=""),
    dict(tag="g_k9s6_perfect", nchr=1, chrlen=800000, k=9, s=6, nreads=120, rlen=100, rep=0.0, opts=""),
    dict(tag="g_k11s2_d20", nchr=2, chrlen=100000, k=11, s=2, nreads=150, rlen=120, rep=0.1, opts="-d 20"),
    dict(tag="g_k20s13_long", nchr=2, chrlen=150000, k=20, s=13, nreads=40, rlen=600, rep=0.1, opts=""),
    dict(tag="g_k13s6_nq", nchr=3, chrlen=100000, k=13, s=6, nreads=200, rlen=150, rep=0.15, opts="-q 5", nfrac=0.02, qualmix=True, varlen=True),
    dict(tag="g_k13s6_x", nchr=2, chrlen=100000, k=13, s=6, nreads=120, rlen=150, rep=0.15, opts="-x"),
    dict(tag="g_k13s3_short", nchr=2, chrlen=150000, k=13, s=3, nreads=200, rlen=36, rep=0.0, opts="", varlen=True),
    dict(tag="g_k13s6_c05", nchr=2, chrlen=100000, k=13, s=6, nreads=150, rlen=150, rep=0.15, opts="-c 0.5", varlen=True),
    # >= 512 reference sequences: concatenated mode (assignSequenceIndex places the alignments, results.c:1695)
    # (post_only: in this mode the reference's RS lines show the alignments AFTER assignSequenceIndex, the stage dumps of the
    #  path show them before -- the fixture serves tests/test_postprocess.py, which is about exactly that step)
    # exact repeat copies (equal best alignments: the choice among them, reads reported for multiple placements) and reads of
    # random bases (unmapped) -- for the report tests (tests/test_report.py)
    dict(tag="g_k13s6_ties", nchr=2, chrlen=80000, k=13, s=6, nreads=220, rlen=100, rep=0.3, opts="", div=0.0, junk=0.08),
    dict(tag="g_k11s4_cat", nchr=600, chrlen=1500, k=11, s=4, nreads=200, rlen=100, rep=0.0, opts="-d -1", qualmix=True, post_only=True, junction=0.3),
]


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def make(cfg, tmp):
    tag = cfg["tag"]
    seed = int(hashlib.md5(tag.encode()).hexdigest()[:6], 16)
    rng = np.random.default_rng(seed + 77)
    extra = {"divergence": cfg["div"]} if "div" in cfg else {}
    ch = synth.make_reference(cfg["nchr"], cfg["chrlen"], seed=seed, repeat_frac=cfg["rep"], n_fam=5, cons_len=300, **extra)
    fa = os.path.join(tmp, tag + ".fa")
    fq = os.path.join(tmp, tag + ".fq")
    synth.write_fasta(fa, ch)
    nfrac = cfg.get("nfrac", 0.0)
    if nfrac > 0:
        lines = open(fa).read().split("\n")
        for i, ln in enumerate(lines):
            if ln and ln[0] != ">" and rng.random() < nfrac:
                p = int(rng.integers(0, len(ln)))
                lines[i] = ln[:p] + "N" * min(5, len(ln) - p) + ln[p + 5:]
        open(fa, "w").write("\n".join(lines))
    reads, _ = synth.make_reads(ch, cfg["nreads"], cfg["rlen"], seed=seed + 1, sub_rate=0.02, indel_read_frac=0.2)
    if cfg.get("junction"):             # reads across the junction of two consecutive sequences (splitMultiSpan, results.c:1472)
        for i in range(len(reads)):
            if rng.random() < cfg["junction"]:
                c = int(rng.integers(0, cfg["nchr"] - 1))
                cut = int(rng.integers(15, cfg["rlen"] - 15))
                r = np.concatenate([ch[c][len(ch[c]) - cut:], ch[c + 1][:cfg["rlen"] - cut]]).copy()
                mut = rng.random(len(r)) < 0.02
                r = np.where(mut, (r + rng.integers(1, 4, size=len(r))) & 3, r).astype(np.uint8)
                if rng.random() < 0.15:      # an indel next to the junction
                    p_ = min(len(r) - 2, max(1, cut + int(rng.integers(-6, 7))))
                    r = np.concatenate([r[:p_], r[p_ + 1:], rng.integers(0, 4, size=1, dtype=np.uint8)])
                reads[i] = synth.revcomp_codes(r) if rng.random() < 0.5 else r
    if cfg.get("junk"):                 # reads that are not from the reference
        for i in range(len(reads)):
            if rng.random() < cfg["junk"]:
                reads[i] = rng.integers(0, 4, size=cfg["rlen"], dtype=np.uint8)
    with open(fq, "wb") as f:
        for i, r in enumerate(reads):
            b = bytearray(synth.codes_to_ascii(r))
            if cfg.get("varlen"):
                b = b[:int(rng.integers(max(8, cfg["k"] - 2), cfg["rlen"] + 1))]
            if nfrac > 0 and rng.random() < 0.3:
                b[int(rng.integers(0, len(b)))] = ord("N")
            if rng.random() < 0.1:
                b = bytearray(bytes(b).lower())
            q = bytearray(b"I" * len(b))
            if cfg.get("qualmix"):
                for j in range(len(b)):
                    if rng.random() < 0.05:
                        q[j] = 33 + int(rng.integers(0, 12))
            f.write(b"@r%d\n" % i + bytes(b) + b"\n+\n" + bytes(q) + b"\n")
    pre = os.path.join(tmp, tag)
    subprocess.run([os.path.join(REF, "smalt"), "index", "-k", str(cfg["k"]), "-s", str(cfg["s"]), pre, fa],
                   check=True, capture_output=True)
    dump = subprocess.run([os.path.join(REF, "refdump")] + cfg["opts"].split() + [pre, fq], check=True,
                          capture_output=True).stdout
    for src, dst in ((fa, tag + ".fa.gz"), (fq, tag + ".fq.gz")):
        with gzip.GzipFile(os.path.join(HERE, dst), "wb", mtime=0) as g:
            g.write(open(src, "rb").read())
    with gzip.GzipFile(os.path.join(HERE, tag + ".refdump.txt.gz"), "wb", mtime=0) as g:
        g.write(dump)
    # the same reads with the state after result post-processing (refdump -p: SURVEY 8f N1), reduced to the lines that matter
    post = subprocess.run([os.path.join(REF, "refdump"), "-n", "-p"] + cfg["opts"].split() + [pre, fq], check=True, capture_output=True).stdout
    keep = [ln for ln in post.split(b"\n") if ln[:2] in (b"RE", b"RC", b"RW", b"RX", b"PS", b"RF", b"SO", b"SS", b"SG")]
    with gzip.GzipFile(os.path.join(HERE, tag + ".post.txt.gz"), "wb", mtime=0) as g:
        g.write(b"\n".join(keep) + b"\n")
    return dict(tag=tag, k=cfg["k"], s=cfg["s"], opts=cfg["opts"], sma_md5=md5(pre + ".sma"), smi_md5=md5(pre + ".smi"),
                dump_lines=dump.count(b"\n"), post_only=bool(cfg.get("post_only")))


if __name__ == "__main__":
    import tempfile
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    with tempfile.TemporaryDirectory() as tmp:
        manifest = [make(c, tmp) for c in CONFIGS]
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
    for m in manifest:
        print(m)
