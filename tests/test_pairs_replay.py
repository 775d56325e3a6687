"""Read pairs above the mapping calls, without a device: the product's own pair logic -- the rounds of
smg_pairrun.hpp with the post-call pass of every call, the proper-pair probe and the search intervals (smg_pairs.hpp), and
smaltgpu_report_emit_pairs (pairing, choice among pairings, mapping qualities, CIGAR / SAM lines of both mates) -- is driven
with the mapping calls the reference recorded for the committed paired fixtures (`refdump -P`, make_golden_pairs.py).
tests/hostemu/pair_check.cpp checks every call the plan asks for against the record (mate, search intervals, threshold,
running maxima) and prints the report; the text must be what the reference PROGRAM printed for the same command line
(make_golden_pair_reports.py), and the pair flags must be the ones rmapPair returned."""
import gzip
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
PAIRS = {e["tag"]: e for e in json.load(open(os.path.join(GOLD, "manifest_pairs.json")))}
REPORTS = [e for e in json.load(open(os.path.join(GOLD, "manifest_pair_reports.json"))) if not e.get("remap")]


@pytest.fixture(scope="session")
def pair_check(tmp_path_factory):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "smalt_amd", "csrc")], check=True)
    exe = str(tmp_path_factory.mktemp("pc") / "pair_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "hostemu", "pair_check.cpp"),
                    "-L" + os.path.join(ROOT, "smalt_amd"), "-lsmaltgpu", "-Wl,-rpath," + os.path.join(ROOT, "smalt_amd"), "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    return exe


def driver_args(opts, k):
    """`smalt map` options -> the keys of pair_check (smalt.c:209-245 formats, :495-504 output flags, menu.c:1487 -r)"""
    a = dict(k=k, dmin=0, dmax=500, lib=1, every=0, fmt=0, mod=0, minsw=18, below=0, minid=0.0, seed=0)
    randrepeat, d_given = True, False
    i = 0
    while i < len(opts):
        o, v = opts[i], opts[i + 1] if i + 1 < len(opts) else None
        if o == "-i":
            a["dmax"] = int(v)
        elif o == "-j":
            a["dmin"] = int(v)
        elif o == "-l":
            a["lib"] = {"pe": 1, "mp": 2, "pp": 3}[v]
        elif o == "-r":
            a["seed"] = int(v)
            randrepeat = int(v) >= 0
        elif o == "-d":
            a["below"] = int(v)
            d_given = True
        elif o == "-y":
            a["minid"] = float(v)
        elif o == "-m":
            a["minsw"] = int(v)
        elif o == "-q":
            pass
        elif o == "-f":
            key, _, mods = v.partition(":")
            if key == "ssaha":
                a["fmt"] = 2
            if key in ("sam", "samsoft"):
                a["fmt"] = 1
                mod = 4 | 2
                for m in mods.split(","):
                    if m == "nohead":
                        mod &= ~4
                    elif m == "clip":
                        mod &= ~2
                    elif m in ("x", "X"):
                        mod |= 8
                a["mod"] = mod
        i += 2
    out = 0
    if a["below"] == 0:
        out |= 1
        if not d_given:
            out |= 2
            if randrepeat:
                out |= 8
    a["out"] = out
    return ["%s=%s" % kv for kv in a.items()]


@pytest.mark.parametrize("rep", REPORTS, ids=["%s-%s" % (e["tag"], e["variant"]) for e in REPORTS])
def test_replayed_pairs_print_what_the_reference_printed(rep, pair_check, tmp_path):
    tag = rep["tag"]
    paths = {}
    for ext in (".fa", "_1.fq", "_2.fq", ".refdump.txt"):
        paths[ext] = str(tmp_path / (tag + ext))
        with gzip.open(os.path.join(GOLD, tag + ext + ".gz"), "rb") as g, open(paths[ext], "wb") as f:
            f.write(g.read())
    seqinfo = str(tmp_path / "seqinfo.txt")
    with open(paths[".fa"]) as f, open(seqinfo, "w") as o:
        name, n = None, 0
        for ln in f:
            if ln.startswith(">"):
                if name:
                    o.write("%s %d\n" % (name, n))
                name, n = ln[1:].split()[0], 0
            else:
                n += len(ln.strip())
        o.write("%s %d\n" % (name, n))
    for threads in (1, 3):
        r = subprocess.run([pair_check, paths[".refdump.txt"], paths["_1.fq"], paths["_2.fq"], seqinfo, "threads=%d" % threads] + driver_args(rep["opts"], PAIRS[tag]["k"]),
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        with gzip.open(os.path.join(GOLD, "%s.%s.out.gz" % (tag, rep["variant"])), "rt") as g:
            want = [ln for ln in g.read().split("\n") if not ln.startswith("@")]
        got = r.stdout.split("\n")
        for i, (x, y) in enumerate(zip(got, want)):
            assert x == y, "line %d" % (i + 1)
        assert len(got) == len(want)
        # pair flags as rmapPair returned them (`PE n err= pairflg= ncalls=` lines of the record)
        # -- except which-mate-first (bit 1) of a pair with a mate shorter than a word: the reference compares the long mate's
        # hits with what the PREVIOUS pair left in the short mate's hit info (collectHitInfo returns before clearing it,
        # hashhit.c:525-526), and nothing reads the bit afterwards
        flags = [(int(ln.split()[2]), int(ln.split()[3])) for ln in r.stderr.split("\n") if ln.startswith("FLG")]
        with open(paths[".refdump.txt"]) as f:
            ref = [int(ln.split("pairflg=")[1].split()[0]) for ln in f if ln.startswith("PE ")]
        assert len(flags) == len(ref)
        for (got_f, short), want_f in zip(flags, ref):
            mask = ~2 if short else ~0
            assert got_f & mask == want_f & mask
