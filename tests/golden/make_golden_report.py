#!/usr/bin/env python3
"""Generates the report fixtures `<tag>.<variant>.out.gz`: what the reference program `smalt map` (oracle/_ref/smalt, the
reference compiled here by oracle/Makefile) prints for the golden inputs -- CIGAR and SAM lines, the selection among equal
best alignments (-r), output filters (-m, -y) -- one file per (fixture, command-line variant) of manifest_report.json.
Data only; needs /root/reference (through oracle/_ref) and is not run by the tests.

    python tests/golden/make_golden_report.py [variant ...]       (variants named: only those are made, the others stay as they are)
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
# variant name -> extra options of `smalt map` (the fixture's own options come first)
VARIANTS = {
    "cigar": ["-r", "3", "-f", "cigar"],
    "sam": ["-r", "3", "-f", "sam"],
    "samx": ["-r", "5", "-f", "sam:nohead,x"],
    "samclip": ["-r", "-1", "-f", "sam:clip"],
    "cigar_norand": ["-r", "-1", "-f", "cigar"],
    "cigar_filt": ["-r", "3", "-m", "60", "-y", "0.9", "-f", "cigar"],
    "cigar_d0": ["-r", "3", "-d", "0", "-f", "cigar"],
    "sam_d0": ["-r", "3", "-d", "0", "-f", "samsoft"],
    "cigar_r7": ["-r", "7", "-f", "cigar"],
    "cigar_dall": ["-r", "3", "-d", "-1", "-f", "cigar"],
    "ssaha": ["-r", "3", "-f", "ssaha"],
    "ssaha_d0": ["-r", "3", "-d", "0", "-f", "ssaha"],
}
# the same reads spelled differently (golden_util.reshape_reads): variant -> (style, options)
RESHAPED = {"wrapped_sam": ("wrapped", ["-r", "3", "-f", "sam"]), "fasta_sam": ("fasta", ["-r", "3", "-f", "sam"]), "fasta_cigar": ("fasta", ["-r", "3", "-f", "cigar"])}
# every fixture gets cigar + sam; the others go to a few fixtures to keep the data small
EXTRA = {"g_k13s6_ties": ["samx", "samclip", "cigar_norand", "cigar_filt", "cigar_d0", "sam_d0", "cigar_r7", "cigar_dall", "ssaha", "ssaha_d0"],
         "g_k13s6_hash": ["samx", "samclip", "cigar_norand", "cigar_filt", "cigar_d0", "ssaha"], "g_k11s2_d20": ["samx", "cigar_filt"],
         "g_k11s4_cat": ["samclip", "cigar_norand", "ssaha"], "g_k13s3_short": ["samx", "cigar_d0", "ssaha"]}


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    only = set(sys.argv[1:])
    mpath = os.path.join(HERE, "manifest_report.json")
    man = [m for m in json.load(open(mpath)) if m["variant"] not in only] if only else []
    with tempfile.TemporaryDirectory() as tmp:
        for e in gu.MANIFEST_ALL:
            fx = gu.unpack(e, tmp)
            for v in ["cigar", "sam"] + EXTRA.get(e["tag"], []):
                if only and v not in only:
                    continue
                opts = e["opts"].split()
                if "-c" in opts and "-x" not in opts:
                    opts = ["-x"] + opts                      # `smalt map` takes -c only together with -x
                if "-d" in VARIANTS[v] and "-d" in opts:
                    continue
                out = os.path.join(tmp, "o.txt")
                cmd = [SMALT, "map"] + opts + VARIANTS[v] + ["-o", out, fx["prefix"], fx["fq"]]
                subprocess.run(cmd, check=True, capture_output=True)
                txt = open(out, "rb").read()
                with gzip.GzipFile(os.path.join(HERE, "%s.%s.out.gz" % (e["tag"], v)), "wb", mtime=0) as g:
                    g.write(txt)
                # a different -d changes the mapping itself: the raw alignments of the fixture do not apply (whole-program test only)
                remap = "-d" in VARIANTS[v] and VARIANTS[v][VARIANTS[v].index("-d") + 1] != "0"
                man.append(dict(tag=e["tag"], variant=v, opts=opts + VARIANTS[v], lines=txt.count(b"\n"), remap=remap))
                print(e["tag"], v, txt.count(b"\n"), "lines")
            if e["tag"] in ("g_k13s6_hash", "g_k13s6_nq") and not only:
                for v, (style, vopts) in RESHAPED.items():
                    if style == "fasta" and "-q" in e["opts"]:
                        continue          # without base qualities the seeding differs: the raw alignments of the fixture do not apply
                    inp = gu.reshape_reads(fx["fq"], style, os.path.join(tmp, "reshaped.txt"))
                    out = os.path.join(tmp, "o.txt")
                    opts = e["opts"].split()
                    subprocess.run([SMALT, "map"] + opts + vopts + ["-o", out, fx["prefix"], inp], check=True, capture_output=True)
                    txt = open(out, "rb").read()
                    with gzip.GzipFile(os.path.join(HERE, "%s.%s.out.gz" % (e["tag"], v)), "wb", mtime=0) as g:
                        g.write(txt)
                    man.append(dict(tag=e["tag"], variant=v, opts=opts + vopts, lines=txt.count(b"\n"), input=style))
                    print(e["tag"], v, txt.count(b"\n"), "lines")
    json.dump(man, open(mpath, "w"), indent=1)


if __name__ == "__main__":
    main()
