"""Replay of the split-read reference dumps (`oracle/_ref/refdump -s -p -n`, tests/golden/make_golden_split.py): per read one
block per mapSingleRead call of rmapSingle under RMAPFLG_SPLIT -- the read's own call and, where the best alignment of the
first read segment leaves room, mapSecondary's call with k-mer words from the uncovered stretch (rmap.c:1435-1505) -- in the
line format of the paired dumps (pair_replay.py), plus the alignment set after each call (`PS`, `RF`, `SO`, `SS`, `SG`) and,
behind the read's `PE` line, as rmapSingle returns it."""
import gzip
import json
import os

import golden_util as gu
import pair_replay as pr

MANIFEST = json.load(open(os.path.join(gu.GOLD, "manifest_split.json")))
POST_TAGS = ("PS", "RF", "SO", "SS", "SG")


def parse(text):
    """-> reads as pair_replay.parse gives pairs, every call with .post (lines of the set after the call), every read with
    .post_final; `RW` lines (the set's alignments ahead of a call's pass) are dropped"""
    plain, posts, key = [], {}, None
    nread, ncall = -1, -1
    for ln in text.split("\n"):
        tag = ln.split(" ", 1)[0]
        if tag == "PAIR":
            nread += 1
            ncall = -1
        elif tag == "MS":
            ncall += 1
        elif tag == "PE":
            ncall = "final"
        if tag in POST_TAGS:
            posts.setdefault((nread, ncall), []).append(ln)
            continue
        if tag == "RW":
            continue
        plain.append(ln)
    reads = pr.parse("\n".join(plain))
    for i, R in enumerate(reads):
        for j, c in enumerate(R["calls"]):
            c["post"] = posts.get((i, j), [])
        R["post_final"] = posts.get((i, "final"), [])
    return reads


def post_state(lines):
    """PS / RF / SO / SS / SG lines -> dict(ps, rows, so, ss, sg)"""
    st = dict(ps=None, rows=[], so=[], ss=None, sg=None)
    for ln in lines:
        f = ln.split()
        if f[0] == "PS":
            st["ps"] = [int(x) for x in f[1:]]
        elif f[0] == "RF":
            st["rows"].append(dict(status=int(f[2]), score=int(f[3]), mapscor=int(f[4]), prob=float(f[5]), q_start=int(f[6]), q_end=int(f[7]), s_start=int(f[8]),
                                   s_end=int(f[9]), sidx=int(f[10]), rsltx=int(f[11]), qsegx=int(f[12]), swrank=int(f[13]), diffstr=bytes.fromhex(f[14]) if len(f) > 14 else b""))
        elif f[0] in ("SO", "SS", "SG"):
            st[f[0].lower()] = [int(x) for x in f[1:]]
    return st


def second_call_range(state, qlen, k, s):
    """mapSecondary's choice (rmap.c:1459-1481) from the set after the first call -> (first, last) or None"""
    if not state["so"] or state["ss"] is None or not state["sg"] or len(state["sg"]) < 2 or state["sg"][1] <= state["sg"][0]:
        return None
    top = state["rows"][state["ss"][state["sg"][0]]]
    lo, hi = top["q_start"], top["q_end"]
    assert lo <= hi <= qlen
    if lo + hi > qlen:
        a, b = 0, (lo - 2 if lo > 1 else 0)
    else:
        a, b = hi, qlen - 1
    if a + k + s > b + 1:
        return None
    return a, b


def load_fixture(entry, tmpdir):
    import oracle_lib as ol
    tag = entry["tag"]
    paths = {}
    for ext in (".fa", ".fq"):
        p = os.path.join(str(tmpdir), tag + ext)
        with gzip.open(os.path.join(gu.GOLD, tag + ext + ".gz"), "rb") as g, open(p, "wb") as f:
            f.write(g.read())
        paths[ext] = p
    names, seqs = gu.read_fasta(paths[".fa"])
    ix = ol.build_index(seqs, names, entry["k"], entry["s"])
    prefix = os.path.join(str(tmpdir), tag)
    assert ol.lib().or_index_write(ix, prefix.encode()) == 0
    ol.lib().or_index_free(ix)
    with gzip.open(os.path.join(gu.GOLD, tag + ".refdump.txt.gz"), "rt") as g:
        reads = parse(g.read())
    opts = entry["opts"].split()
    return dict(prefix=prefix, reads=gu.read_fastq(paths[".fq"]), dump=reads, min_basq=int(opts[opts.index("-q") + 1]) if "-q" in opts else 0,
                names=names, seqs=seqs, fq=paths[".fq"])


def planned_calls(fx, entry):
    """every recorded call with the arguments a caller derives for it: (read number, call, seed_range or None)"""
    out = []
    for R in fx["dump"]:
        qlen = len(fx["reads"][R["no"]][1])
        for j, c in enumerate(R["calls"]):
            rng = None
            if j == 1:
                rng = second_call_range(post_state(R["calls"][0]["post"]), qlen, entry["k"], entry["s"])
                assert rng is not None, "the reference made a second call where the rule names no stretch (read %d)" % R["no"]
            assert j < 2
            out.append((R["no"], c, rng))
        if len(R["calls"]) == 1:
            st = post_state(R["calls"][0]["post"])
            assert second_call_range(st, qlen, entry["k"], entry["s"]) is None or R["calls"][0]["err"], "the rule names a stretch, the reference made no second call (read %d)" % R["no"]
    return out
