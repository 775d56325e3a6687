"""Known-answer vectors for cutting an alignment string to a window of reference offsets (the operation behind
splitMultiSpan, results.c:1472): random alignment strings and windows go through the REFERENCE's own diffStrSegment
(diffstr.c:1370, from oracle/_ref/libsmaltref.so, which oracle/Makefile compiles from the reference's sources where they
lie); inputs and outputs are stored as data in kat_diffstr_cut.json.gz.  Run in the container that has /root/reference:

    python tests/golden/make_golden_cut.py

tests/test_post_cut.py replays them through smgpost::cut_window (smalt_amd/csrc/smg_post.hpp)."""
import ctypes as C
import gzip
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


class DiffStr(C.Structure):
    _fields_ = [("dstrp", C.POINTER(C.c_ubyte)), ("len", C.c_int), ("n_alloc", C.c_int), ("blksz", C.c_int)]


def encode(cols):
    """columns 'm' (match) 's' (substitution) 'i' 'd' -> bytes as the reference's aligner writes them: at most 61 matched
    bases in front of an event, a full MATCH byte (62 bases) when a run goes on, a closing S byte and the terminator"""
    out, run = bytearray(), 0
    OP = {"d": 1, "i": 2, "s": 3}
    for c in cols:
        if c == "m":
            run += 1
            if run == 62:
                out.append(61)
                run = 0
        else:
            out.append(OP[c] << 6 | run)
            run = 0
    out.append(3 << 6 | run)
    out.append(0)
    return bytes(out)


def random_columns(rng):
    n = rng.choice([5, 20, 60, 150, 400])
    p_event = rng.choice([0.0, 0.02, 0.1, 0.4])
    cols = []
    while len(cols) < n:
        if rng.random() < p_event:
            kind = rng.choice("sssiidd")
            cols.extend(kind * (1 if kind == "s" or rng.random() < 0.6 else rng.randint(2, 5)))
        else:
            cols.append("m")
    if rng.random() < 0.8:
        cols[0] = "m"
    if rng.random() < 0.8:
        cols[-1] = "m"
    return cols


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsmaltref.so"))
    lib.diffStrCreate.restype = C.POINTER(DiffStr)
    lib.diffStrCreate.argtypes = [C.c_int]
    lib.diffStrSegment.argtypes = [C.POINTER(DiffStr), C.c_char_p, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4
    buf = lib.diffStrCreate(256)
    rng = random.Random(20261004)
    vec = []
    for _ in range(6000):
        cols = random_columns(rng)
        s = encode(cols)
        nref = sum(1 for c in cols if c != "i")
        for _w in range(4):
            lo = rng.randint(0, max(0, nref - 1))
            hi = rng.randint(lo, nref + 3) if rng.random() < 0.9 else rng.randint(0, nref)
            o = [C.c_int(-1) for _ in range(4)]
            rv = lib.diffStrSegment(buf, s, lo, hi, *[C.byref(x) for x in o])
            rec = {"s": s.hex(), "lo": lo, "hi": hi, "rv": rv}
            if rv == 0:
                rec["out"] = bytes(buf.contents.dstrp[i] for i in range(buf.contents.len)).hex()
                rec["ref"] = [o[0].value, o[1].value]
                rec["read"] = [o[2].value, o[3].value]
            vec.append(rec)
    with gzip.open(os.path.join(HERE, "kat_diffstr_cut.json.gz"), "wt") as g:
        json.dump(vec, g)
    codes = {}
    for r in vec:
        codes[r["rv"]] = codes.get(r["rv"], 0) + 1
    print("wrote %d vectors, return codes %s" % (len(vec), codes))


if __name__ == "__main__":
    main()
