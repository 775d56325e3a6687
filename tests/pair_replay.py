"""Replay of the paired-mode reference dumps (`oracle/_ref/refdump -P`, oracle/DUMPFORMAT.md): the reference's rmapPair
(rmap.c:1744) makes two to four mapSingleRead calls per pair -- rare mate, interval-restricted mate, unrestricted re-map,
restricted re-map over the on-the-fly k=5 index -- and the dump records every call with its arguments (`MS`, `IV`, `PL`
lines) and its stage state.  A mapper under test (the CPU oracle or the GPU library) is fed each call's arguments and must
reproduce the call's block.

What a call adds to a ResultSet that already holds alignments follows resultSetAddFromAli (results.c:1852-1942): the call
returns every alignment of its traceback pass and `append_rule` puts them behind the set with the reference's handling of
repeats (an alignment equal to the one before it is taken off again, the alignment behind it is lost); the set's running score maxima go into the call (the traceback pass raises its
threshold to the set's second-best score, rmap.c:881-885) and come out updated."""


def parse(text):
    pairs, cur, call = [], None, None
    for ln in text.split("\n"):
        if not ln:
            continue
        tag = ln.split(" ", 1)[0]
        if tag == "PAIR":
            f = ln.split()
            cur = dict(no=int(f[1]), names=(f[2], f[3]), calls=[], pe=None)
            pairs.append(cur)
        elif tag == "MS":
            kv = dict(x.split("=") for x in ln.split()[2:])
            call = dict(mate=int(kv["mate"]), niv=int(kv["niv"]), fine=int(kv["fine"]), minscor=int(kv["minscor"]), mincov=int(kv["mincov"]),
                        belowmax=int(kv["belowmax"]), flags=int(kv["flags"]), prevmax=tuple(int(x) for x in kv["prevmax"].split(",")),
                        ivs=[], pl=None, lines=[], rs=[], rx=None, err=None)
            cur["calls"].append(call)
        elif tag == "IV":
            f = ln.split()
            call["ivs"].append((int(f[1]), int(f[2]), int(f[3])))
        elif tag == "PL":
            call["pl"] = result_of(ln)
        elif tag == "PE":
            cur["pe"] = ln
            call = None
        elif tag == "READ":
            call["err"] = int(ln.rsplit("err=", 1)[1])
            call["lines"].append(ln.rsplit(" err=", 1)[0])
        elif tag == "RS":
            call["rs"].append(result_of(ln))
        elif tag == "RX":
            call["rx"] = tuple(int(x) for x in ln.split()[1:])
        else:
            call["lines"].append(ln)
    return pairs


def result_of(ln):
    f = ln.split()
    return dict(reverse=1 if f[2] == "R" else 0, score=int(f[3]), q_start=int(f[4]), q_end=int(f[5]), s_start=int(f[6]), s_end=int(f[7]),
                sidx=int(f[8]), diffstr=bytes.fromhex(f[9]) if len(f) > 9 else b"")


def same_alignment(a, b):                   # isIdenticalResult, results.c:556-565
    return all(a[k] == b[k] for k in ("s_start", "s_end", "q_start", "q_end", "score", "sidx"))


def append_rule(results, cand_first, prev_last, prev_count=None):
    """What a call's RAW alignments (every alignment the traceback pass produced, candidate by candidate) leave behind a
    ResultSet whose last alignment is prev_last: resultSetAddFromAli (results.c:1852-1942) once per candidate, as a machine over
    the array's physical slots.  A candidate opens a slot; every alignment is written to the open slot and compared with the slot
    before it; an alignment that repeats it gives the slot back (the array shrinks by one, the slot stays open and outside the
    array), anything else stays and the next alignment opens the slot behind the array's end -- which is the same slot again
    after a repeat, so the alignment that follows a repeat is overwritten (or left outside the array at the end of the
    candidate): it is lost, although it went through the score maxima.  -> the alignments added (the array behind the old end)"""
    slots = [prev_last] if prev_last is not None else []
    if prev_last is not None and prev_count is not None and prev_count > 1:
        slots = [None] * (prev_count - 1) + [prev_last]
    n_old = length = len(slots)
    i = 0
    while i < len(results):
        j = i + 1
        while j < len(results) and not cand_first[j]:
            j += 1
        s = length                       # the candidate's first slot
        length += 1
        fresh = False
        for a in results[i:j]:
            if fresh:
                s = length
                length += 1
                fresh = False
            if s == len(slots):
                slots.append(None)
            slots[s] = a
            fresh = length < 2 or slots[s - 1] is None or not same_alignment(a, slots[s - 1])
            if not fresh:
                length -= 1
        i = j
    assert length >= n_old, "the call took an older alignment off the set"
    return slots[n_old:length]


def stage_lines(dump_text):
    """(stage lines without RS/RX, err stripped from the READ line) of one dump block"""
    out = []
    for ln in dump_text.split("\n"):
        if not ln or ln[:2] in ("RS", "RX", "HL"):
            continue
        out.append(ln.rsplit(" err=", 1)[0] if ln.startswith("READ") else ln)
    return out


def check_call(call, produced_lines, results, cand_first, stats):
    """stats: (swmax, sw2nd, nseg, nseg_tot, nhit, nhit_tot) of the produced call, which was given call["prevmax"]"""
    exp = call["lines"]
    for i, (x, y) in enumerate(zip(produced_lines, exp)):
        assert x == y, ("stage line %d" % i, x, y)
    assert len(produced_lines) == len(exp), (len(produced_lines), len(exp))
    got = append_rule(results, cand_first, call["pl"])
    assert len(got) == len(call["rs"]), (len(got), len(call["rs"]))
    for a, b in zip(got, call["rs"]):
        assert a == b, (a, b)
    assert (len(got),) + tuple(stats[0:6]) == call["rx"], ((len(got),) + tuple(stats[0:6]), call["rx"])


def load_fixture(entry, tmpdir):
    """-> dict(prefix (index files written by the oracle), reads1, reads2 [(name, seq, qual)], pairs (parsed dump), min_basq)"""
    import gzip
    import os

    import golden_util as gu
    import oracle_lib as ol
    tag = entry["tag"]
    paths = {}
    for ext in ("fa", "_1.fq", "_2.fq"):
        p = os.path.join(str(tmpdir), tag + (ext if ext[0] == "_" else "." + ext))
        with gzip.open(os.path.join(gu.GOLD, tag + (ext if ext[0] == "_" else "." + ext) + ".gz"), "rb") as g, open(p, "wb") as f:
            f.write(g.read())
        paths[ext] = p
    names, seqs = gu.read_fasta(paths["fa"])
    ix = ol.build_index(seqs, names, entry["k"], entry["s"])
    prefix = os.path.join(str(tmpdir), tag)
    assert ol.lib().or_index_write(ix, prefix.encode()) == 0
    ol.lib().or_index_free(ix)
    with gzip.open(os.path.join(gu.GOLD, tag + ".refdump.txt.gz"), "rt") as g:
        pairs = parse(g.read())
    opts = entry["opts"].split()
    min_basq = int(opts[opts.index("-q") + 1]) if "-q" in opts else 0
    return dict(prefix=prefix, reads1=gu.read_fastq(paths["_1.fq"]), reads2=gu.read_fastq(paths["_2.fq"]), pairs=pairs, min_basq=min_basq,
                names=names, seqs=seqs)
