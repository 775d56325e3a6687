"""BASELINE.json configs[4] at full size -- 8 kbp PacBio-shape reads (3 % substitutions, 5 % insertions, 4 % deletions)
against the 3 Gbp reference (24 x 125 Mbp), k=20 s=13: collision-type index built by the library, reads through the C ABI,
checked by size-independent properties (the oracle needs minutes per read here):
  * truth recovery -- the best alignment of a read lies on its source locus and strand and spans most of the read
  * score bounds   -- every reported DiffStr re-scores to its reported score and consumes exactly the reported read and
                      reference intervals
  * batch invariance -- the same reads mapped as one batch and one at a time give identical results
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K, S, NCHR, CHRLEN, RLEN, NREADS = 20, 13, 24, 125_000_000, 8000, 12


def _walk(d):
    """(score, read bases, reference bases) consumed by a DiffStr (diffstr.h:29-75)."""
    score = q = r = 0
    prev = 0
    for i, b in enumerate(d):
        if b == 0:
            break
        typ, m = b >> 6, b & 0x3F
        last = i + 1 >= len(d) or d[i + 1] == 0
        score += m
        q += m
        r += m
        if typ == 0:
            score += 1; q += 1; r += 1; prev = 0
        elif typ == 3:
            if not last:
                score -= 2; q += 1; r += 1
            prev = 0
        else:
            score += -3 if (prev == typ and m == 0) else -4
            prev = typ
            if typ == 1:
                r += 1          # deletion from the read: a reference base without a read base
            else:
                q += 1
    return score, q, r


def test_long_reads_against_the_full_size_reference():
    import torch
    from smalt_amd import api, gpuindex, synth
    dev = torch.device("cuda", 0)
    sop = [i * CHRLEN for i in range(NCHR + 1)]
    ref = gpuindex.make_reference_gpu(NCHR, CHRLEN, 20261004, dev)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    ascii_ref = torch.cat([lut[c.long()] for c in ref.split(1 << 28)])
    gix = api.Index.build_device(ascii_ref.data_ptr(), sop, ["chr%d" % (i + 1) for i in range(NCHR)], K, S, 0)
    del ascii_ref
    assert gix.info().typ == 1                  # 4^20 keys do not fit: hash32mix (smalt.c:298-328)
    rng = np.random.default_rng(99)
    reads, truth = [], []
    for i in range(NREADS):
        c, p = int(rng.integers(0, NCHR)), int(rng.integers(0, CHRLEN - RLEN - 16))
        src = ref[c * CHRLEN + p: c * CHRLEN + p + RLEN + 8].cpu().numpy()
        r, t = synth.make_long_reads([src], 1, RLEN, seed=500 + i)
        reads.append(synth.codes_to_ascii(r[0]))
        truth.append((c, p + int(t[0][1]), int(t[0][2])))
    del ref
    torch.cuda.empty_cache()
    mp = api.Mapper(gix, NREADS, max(len(r) for r in reads))
    par = gix.default_params()
    try:
        res, stats = mp.map_batch(reads, [b"5" * len(r) for r in reads], par)
        ok = 0
        for i, rr in enumerate(res):
            assert stats[i]["err"] == 0 and rr, i
            for a in rr:
                sc, q, r = _walk(a["diffstr"])
                assert sc == a["score"], (i, sc, a["score"])
                assert q == a["q_end"] - a["q_start"] + 1 and r == a["s_end"] - a["s_start"] + 1, i
                assert 1 <= a["score"] <= len(reads[i])
            b = max(rr, key=lambda a: a["score"])
            c, p, strand = truth[i]
            if b["sidx"] == c and b["reverse"] == strand and abs(int(b["s_start"]) - 1 - p) < 400 and b["q_end"] - b["q_start"] > 0.9 * len(reads[i]):
                ok += 1
        assert ok >= NREADS - 1, ok
        for i in (0, 5):                           # one read per batch: identical results
            r1, s1 = mp.map_batch([reads[i]], [b"5" * len(reads[i])], par)
            assert r1[0] == res[i] and s1[0] == stats[i]
    finally:
        mp.close()
        gix.close()
