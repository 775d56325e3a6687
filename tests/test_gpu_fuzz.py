"""Randomised differential test of the whole path against the CPU oracle: seeded references and reads across word
lengths, strides, index types, score penalties, thresholds, depth limits, exhaustive mode, read lengths from below k
to 700 bases, non-ACGT bases and low base qualities.  Every read: identical raw result array and per-read scalars, or an
error for the reads the reference fails."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

FLG_BEST, FLG_SEQBYSEQ, FLG_SENSITIVE = 0x02, 0x10, 0x80

CASES = [
    # k, s, nseq, seqlen, read length range, parameter overrides
    dict(k=13, s=6, nseq=3, seqlen=400_000, rl=(30, 260), par={}),
    dict(k=13, s=6, nseq=3, seqlen=400_000, rl=(100, 101), par=dict(match=2, mismatch=-3, gap_init=-5, gap_ext=-2)),
    dict(k=11, s=2, nseq=2, seqlen=150_000, rl=(25, 120), par=dict(target_depth=20, max_depth=60)),
    dict(k=13, s=3, nseq=4, seqlen=200_000, rl=(11, 80), par={}),
    dict(k=20, s=13, nseq=2, seqlen=500_000, rl=(200, 700), par={}),
    dict(k=13, s=6, nseq=3, seqlen=400_000, rl=(60, 200), par=dict(exhaustive=True, ncut=50)),
    dict(k=13, s=6, nseq=3, seqlen=400_000, rl=(60, 200), par=dict(min_swatscor=60, below_max=12, best=False)),
    dict(k=9, s=6, nseq=5, seqlen=60_000, rl=(20, 150), par=dict(min_basq=10)),
    dict(k=13, s=6, nseq=600, seqlen=3_000, rl=(50, 150), par={}),                       # concatenated mode (>= 512 sequences)
    dict(k=13, s=13, nseq=2, seqlen=400_000, rl=(80, 180), par=dict(below_max=-1, best=False)),
    dict(k=17, s=8, nseq=3, seqlen=300_000, rl=(255, 257), par={}),                      # lengths around the two candidate kernels
    dict(k=13, s=6, nseq=3, seqlen=400_000, rl=(60, 160), par=dict(cov_frac=0.3)),
    dict(k=13, s=6, nseq=2, seqlen=300_000, rl=(250, 2500), par={}),                     # long reads, many hits
    dict(k=13, s=6, nseq=3, seqlen=400_000, rl=(60, 160), par=dict(cov_frac=0.7)),       # min_ktup > 1: sequential candidate stage
    # K3's row form scans the horizontal gap score, which is exact for gap open >= gap extension: the boundary case ...
    dict(k=13, s=6, nseq=3, seqlen=400_000, rl=(60, 160), par=dict(match=1, mismatch=-2, gap_init=-3, gap_ext=-3)),
    # ... and penalties on the other side of it, which keep the anti-diagonal form
    dict(k=13, s=6, nseq=3, seqlen=400_000, rl=(60, 160), par=dict(match=2, mismatch=-3, gap_init=-2, gap_ext=-4)),
    dict(k=13, s=6, nseq=3, seqlen=400_000, rl=(40, 250), par=dict(match=3, mismatch=-4, gap_init=-8, gap_ext=-1, best=False, below_max=30)),
]


def _make(case, seed):
    from smalt_amd import synth
    rng = np.random.default_rng(seed)
    ch = synth.make_reference(case["nseq"], case["seqlen"], seed=seed, repeat_frac=0.2, n_fam=3, cons_len=min(400, case["seqlen"] // 4), divergence=0.06)
    seqs = [bytearray(synth.codes_to_ascii(c)) for c in ch]
    for sq in seqs[: min(3, len(seqs))]:                       # a run of N in the reference
        p = int(rng.integers(0, len(sq) - 300)); sq[p:p + 40] = b"N" * 40
    seqs = [bytes(x) for x in seqs]
    reads, quals = [], []
    lo, hi = case["rl"]
    for i in range(260):
        c = int(rng.integers(0, len(seqs))); ln = int(rng.integers(lo, hi + 1))
        ln = min(ln, len(seqs[c]) - 2)
        p = int(rng.integers(0, len(seqs[c]) - ln - 1))
        r = bytearray(seqs[c][p:p + ln])
        mode = i % 8
        out = bytearray()
        for ch_ in r:
            u = rng.random()
            if mode == 7 and u < 0.06:
                continue
            if mode == 6 and u < 0.05:
                out.append(ch_); out.append(b"ACGT"[int(rng.integers(0, 4))]); continue
            if u < (0.0 if mode == 0 else 0.02 + 0.01 * mode):
                out.append(b"ACGT"[int(rng.integers(0, 4))])
            else:
                out.append(ch_)
        if mode == 3 and len(out) > 12:
            out[int(rng.integers(0, len(out)))] = ord("N")
        if mode == 5:                                           # unrelated read
            out = bytearray(b"ACGT"[int(x)] for x in rng.integers(0, 4, len(out)))
        r = bytes(out) if out else b"A"
        if rng.random() < 0.5:
            r = r[::-1].translate(bytes.maketrans(b"ACGTN", b"TGCAN"))
        q = bytearray(b"I" * len(r))
        if case["par"].get("min_basq"):
            for j in range(len(q)):
                if rng.random() < 0.08:
                    q[j] = 33 + int(rng.integers(0, 12))
        reads.append(r); quals.append(bytes(q))
    return seqs, reads, quals


# SMALT_FUZZ_SEEDS="1000,2000": extra seeds for a stress run (python -m pytest tests/test_gpu_fuzz.py -m gpu)
_EXTRA = [int(x) for x in __import__("os").environ.get("SMALT_FUZZ_SEEDS", "").split(",") if x.strip()]


@pytest.mark.parametrize("seedoff", [0, 77, 154] + _EXTRA, ids=["seedA", "seedB", "seedC"] + ["seed%d" % x for x in _EXTRA])
@pytest.mark.parametrize("ci", range(len(CASES)), ids=["c%d-k%ds%d" % (i, c["k"], c["s"]) for i, c in enumerate(CASES)])
def test_random_workload_matches_oracle(ci, seedoff, oracle_built, tmp_path):
    from smalt_amd import api
    case = CASES[ci]
    seqs, reads, quals = _make(case, 9000 + ci + seedoff)
    names = ["s%d" % i for i in range(len(seqs))]
    oix0 = ol.build_index(seqs, names, case["k"], case["s"])
    pre = str(tmp_path / "fz")
    assert ol.lib().or_index_write(oix0, pre.encode()) == 0
    ol.lib().or_index_free(oix0)
    oix = ol.lib().or_index_read(pre.encode())
    gix = api.Index.load(pre, 0)
    op, gp = ol.default_params(oix), gix.default_params()
    par = case["par"]
    for name_o, name_g, key in (("match", "match", "match"), ("mismatch", "mismatch", "mismatch"), ("gap_init", "gap_init", "gap_init"),
                                ("gap_ext", "gap_ext", "gap_ext"), ("target_depth", "target_depth", "target_depth"), ("max_depth", "max_depth", "max_depth"),
                                ("min_swatscor", "min_swatscor", "min_swatscor"), ("min_swatscor_below_max", "min_swatscor_below_max", "below_max"),
                                ("ncut", "ktuple_maxhit", "ncut"), ("min_basq", "min_basqval", "min_basq")):
        if key in par:
            setattr(op, name_o, par[key]); setattr(gp, name_g, par[key])
    if par.get("best") is False:
        op.flags &= ~FLG_BEST; gp.rmapflg &= ~FLG_BEST
    if par.get("exhaustive"):
        op.flags |= FLG_SENSITIVE; gp.rmapflg |= FLG_SENSITIVE
    om = ol.Mapper(oix)
    exp = []
    for r, q in zip(reads, quals):
        if "cov_frac" in par:                                   # -c below 1.01: a fraction of the read length (smalt.c:1113-1122)
            op.min_cover = min(len(r), int(par["cov_frac"] * len(r)))
        rv, res = om.map(r, q, op)
        st = om.stats()
        # rv != 0: the reference itself fails the read (e.g. ERRCODE_SWATSCOR, alignment.c:768: the traceback's score does
        # not add up to the band pass's maximum -- possible with some penalty sets); the GPU path must flag the same read
        exp.append(((None, rv) if rv else res, dict(swmax=st[0], sw2nd=st[1], nseg=st[2], nseg_tot=st[3], nhit=st[4], nhit_tot=st[5])))
    om.close()
    if "cov_frac" in par:
        gp.min_cover_frac = par["cov_frac"]
    mp = api.Mapper(gix, len(reads), max(len(r) for r in reads))
    try:
        res, stats = mp.map_batch(reads, quals, gp, allow_read_errors=True)
    finally:
        mp.close()
        gix.close()
    nmapped = 0
    for i in range(len(reads)):
        if isinstance(exp[i][0], tuple):
            assert stats[i]["err"] != 0 and res[i] == [], i
            if exp[i][0][1] == 4:               # OR_ERR_SWATSCOR <-> SMALTGPU_ESCORE: the reference's score check keeps its own code
                assert stats[i]["err"] == -8, (i, stats[i]["err"])
            continue
        assert stats[i]["err"] == 0, i
        assert res[i] == exp[i][0], (i, len(reads[i]))
        for kk, v in exp[i][1].items():
            assert stats[i][kk] == v, (i, kk)
        nmapped += 1 if res[i] else 0
    assert nmapped > 100
