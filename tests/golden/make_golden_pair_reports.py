#!/usr/bin/env python3
"""Generates the PAIRED report fixtures `gp_*.<variant>.out.gz`: what the reference program (oracle/_ref/smalt, compiled
here from the reference's sources by oracle/Makefile) prints for the paired golden inputs of make_golden_pairs.py -- CIGAR
lines with pair classes (A/B/C/D/S/R/N), SAM lines with flags, mate fields and template lengths; random, single and full
reporting of ambiguous pairs (-r <seed> / -r -1 / default).  One file per (fixture, variant) of manifest_pair_reports.json.
Data only; needs /root/reference (through oracle/_ref) and is not run by the tests.

    python tests/golden/make_golden_pair_reports.py [variant ...]    (variants named: only those are made, the others stay as they are)
"""
import gzip
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu  # noqa: E402

SMALT = os.path.join(ROOT, "oracle", "_ref", "smalt")
VARIANTS = {
    "cigar": ["-r", "3", "-f", "cigar"],
    "sam": ["-r", "3", "-f", "sam"],
    "samx": ["-r", "5", "-f", "sam:nohead,x"],
    "samclip": ["-r", "-1", "-f", "sam:clip"],
    "cigar_norand": ["-r", "-1", "-f", "cigar"],
    "cigar_all": ["-d", "0", "-f", "cigar"],            # -d given: every equally good pairing is printed
    "sam_all": ["-d", "0", "-f", "sam:nohead"],
    "cigar_ident": ["-r", "3", "-y", "0.95", "-f", "cigar"],
    "cigar_filt": ["-r", "3", "-m", "50", "-y", "0.9", "-f", "cigar"],     # -m changes the mapping itself: whole-program test only
    "ssaha": ["-r", "3", "-f", "ssaha"],
    "ssaha_all": ["-d", "0", "-f", "ssaha"],
}
REMAP = {"cigar_filt"}


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    pairs = json.load(open(os.path.join(HERE, "manifest_pairs.json")))
    only = set(sys.argv[1:])
    mpath = os.path.join(HERE, "manifest_pair_reports.json")
    man = [m for m in json.load(open(mpath)) if m["variant"] not in only] if only else []
    with tempfile.TemporaryDirectory() as tmp:
        for e in pairs:
            tag = e["tag"]
            paths = {}
            for ext in (".fa", "_1.fq", "_2.fq"):
                p = os.path.join(tmp, tag + ext)
                with gzip.open(os.path.join(HERE, tag + ext + ".gz"), "rb") as g, open(p, "wb") as f:
                    f.write(g.read())
                paths[ext] = p
            pre = os.path.join(tmp, tag)
            subprocess.run([SMALT, "index", "-k", str(e["k"]), "-s", str(e["s"]), pre, paths[".fa"]], check=True, capture_output=True)
            for v, vopts in VARIANTS.items():
                if only and v not in only:
                    continue
                if "-d" in e["opts"].split() and "-d" in vopts:      # the fixture's calls were recorded with its own -d (it decides the best-only flag of the mapping)
                    continue
                out = os.path.join(tmp, "o.txt")
                opts = e["opts"].split() + vopts
                subprocess.run([SMALT, "map"] + opts + ["-o", out, pre, paths["_1.fq"], paths["_2.fq"]], check=True, capture_output=True)
                txt = open(out, "rb").read()
                with gzip.GzipFile(os.path.join(HERE, "%s.%s.out.gz" % (tag, v)), "wb", mtime=0) as g:
                    g.write(txt)
                man.append(dict(tag=tag, variant=v, opts=opts, lines=txt.count(b"\n"), remap=v in REMAP))
                print(tag, v, txt.count(b"\n"), "lines")
    json.dump(man, open(mpath, "w"), indent=1)


if __name__ == "__main__":
    main()
