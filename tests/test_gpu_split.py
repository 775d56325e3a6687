"""Split reads (`smalt map -p`: rmapSingle with RMAPFLG_SPLIT, rmap.c:1716-1728 -> mapSecondary, rmap.c:1435-1505) through the
library: smaltgpu_map_split runs both calls of every read as two device batches with the post-call pass between and behind
them, smaltgpu_report_emit prints the partial alignments (class P / SAM flag 0x100, results.c:2250-2278).
  * whole program: `smaltgpu-map -p` prints what the reference program `smalt map -p` (oracle/_ref/smalt) prints, byte for
    byte, for chimeric reads (two or three stretches from different places and strands, substitutions, indels, N, ragged
    lengths) -- CIGAR, SAM and SSAHA lines, sequence-by-sequence and concatenated references, -q, -m/-y filters;
  * the second call on its own: smaltgpu_map_batch_ctx with seed_range / prev_max / raw_alignments against the CPU oracle
    with the same arguments (stage dump lines, alignments, counters)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALT = os.path.join(ROOT, "oracle", "_ref", "smalt")
PROG = os.path.join(ROOT, "smalt_amd", "smaltgpu-map")


def chimeric_reads(ch, nreads, rlen, seed):
    """-> list of (name, bases as bytes): a third plain reads, the others two or three stretches from different places"""
    from smalt_amd import synth
    rng = np.random.default_rng(seed)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    out = []
    for i in range(nreads):
        nparts = 1 if i % 3 == 0 else (3 if i % 10 == 1 else 2)
        cuts = sorted(int(x) for x in rng.integers(25, rlen - 25, size=nparts - 1)) if nparts > 1 else []
        lens = [b - a for a, b in zip([0] + cuts, cuts + [rlen])]
        parts = []
        for ln in lens:
            c = int(rng.integers(0, len(ch)))
            p = int(rng.integers(0, len(ch[c]) - ln - 1))
            s = bytearray(synth.codes_to_ascii(ch[c][p:p + ln]))
            for j in range(len(s)):
                if rng.random() < 0.02:
                    s[j] = b"ACGT"[int(rng.integers(0, 4))]
            if rng.random() < 0.15 and len(s) > 20:
                j = int(rng.integers(5, len(s) - 5))
                if rng.random() < 0.5:
                    del s[j]
                else:
                    s.insert(j, b"ACGT"[int(rng.integers(0, 4))])
            s = bytes(s)
            if rng.random() < 0.5:
                s = s.translate(comp)[::-1]
            parts.append(s)
        b = bytearray(b"".join(parts))
        if i % 9 == 0:
            b[int(rng.integers(0, len(b)))] = ord("N")
        if i % 13 == 0:
            b = b[:int(rng.integers(10, len(b)))]
        out.append(("r%d" % i, bytes(b)))
    return out


def write_inputs(tmp, nchr, chrlen, nreads, rlen, seed):
    from smalt_amd import synth
    ch = synth.make_reference(nchr, chrlen, seed=seed, repeat_frac=0.1, n_fam=3, cons_len=300)
    fa, fq = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fq")
    synth.write_fasta(fa, ch)
    reads = chimeric_reads(ch, nreads, rlen, seed + 1)
    rng = np.random.default_rng(seed + 2)
    with open(fq, "wb") as f:
        for nm, b in reads:
            q = bytes(33 + int(x) for x in rng.integers(5, 41, size=len(b)))
            f.write(b"@" + nm.encode() + b"\n" + b + b"\n+\n" + q + b"\n")
    return ch, fa, fq, reads


@pytest.mark.skipif(not os.path.exists(SMALT), reason="reference binary not built (make -C oracle ref)")
@pytest.mark.parametrize("k,s,nchr,chrlen,rlen,opts", [
    (13, 6, 3, 300_000, 150, ["-f", "cigar"]),
    (11, 3, 2, 200_000, 120, ["-f", "sam", "-q", "10"]),
    (13, 2, 3, 200_000, 200, ["-f", "ssaha", "-r", "-1"]),
    (13, 6, 3, 300_000, 150, ["-f", "sam:nohead,x", "-m", "30", "-y", "0.2"]),
    (13, 6, 600, 2_000, 100, ["-f", "cigar"]),                                   # >= 512 reference sequences: concatenated mode
    (13, 6, 3, 300_000, 250, ["-f", "cigar", "-S", "match=2,subst=-3,gapopen=-5,gapext=-3"]),
])
def test_smaltgpu_map_prints_what_smalt_map_prints_for_split_reads(k, s, nchr, chrlen, rlen, opts, tmp_path):
    tmp = str(tmp_path)
    if "-r" not in opts:
        opts = opts + ["-r", "3"]
    _, fa, fq, reads = write_inputs(tmp, nchr, chrlen, 1500, rlen, seed=k * 1000 + s * 10 + nchr)
    pre = os.path.join(tmp, "idx")
    subprocess.run([SMALT, "index", "-k", str(k), "-s", str(s), pre, fa], check=True, capture_output=True)
    out_ref, out_gpu = os.path.join(tmp, "ref.out"), os.path.join(tmp, "gpu.out")
    subprocess.run([SMALT, "map", "-p"] + opts + ["-o", out_ref, pre, fq], check=True, capture_output=True)
    a = [ln for ln in open(out_ref).read().split("\n") if not ln.startswith("@PG")]
    partial = sum(1 for ln in a if ln.startswith("cigar:P") or ln.startswith("alignment:P") or (len(ln.split("\t")) > 1 and ln.split("\t")[1].isdigit() and int(ln.split("\t")[1]) & 0x100))
    assert partial >= 300, partial            # the inputs do exercise the second call
    for extra in (["-B", "400"], ["-B", "4000", "-n", "1"]):
        r = subprocess.run([PROG, "-p"] + opts + extra + ["-o", out_gpu, pre, fq], capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        b = [ln for ln in open(out_gpu).read().split("\n") if not ln.startswith("@PG")]
        diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
        assert not diff and len(a) == len(b), (len(a), len(b), diff[:3])
    # negative control: without -p the output is a different one (the comparison above is not blind to the partial alignments)
    r = subprocess.run([PROG] + opts + ["-o", out_gpu, pre, fq], capture_output=True)
    assert r.returncode == 0 and open(out_gpu).read().split("\n") != open(out_ref).read().split("\n")


def test_second_call_of_a_split_read_matches_the_oracle(oracle_built, tmp_path):
    """seed_range on its own: words from a stretch of the read only (collectHitInfo with a range, hashhit.c:536-551), appended to
    a set with running score maxima -- device against the CPU restatement, stage state line by line"""
    from smalt_amd import api
    import oracle_lib as ol
    import pair_replay as pr
    tmp = str(tmp_path)
    ch, fa, fq, reads = write_inputs(tmp, 3, 200_000, 400, 150, seed=4711)
    from smalt_amd import synth
    seqs = [synth.codes_to_ascii(c) for c in ch]
    names = ["chr%d" % (i + 1) for i in range(len(seqs))]
    oix = ol.build_index(seqs, names, 13, 4)
    prefix = os.path.join(tmp, "ix")
    assert ol.lib().or_index_write(oix, prefix.encode()) == 0
    gix = api.Index.load(prefix, 0)
    rng = np.random.default_rng(99)
    rd = [(nm, b, bytes(33 + int(x) for x in rng.integers(5, 41, size=len(b)))) for nm, b in reads if len(b) >= 40]
    ranges, prevmax = [], []
    for i, (nm, b, q) in enumerate(rd):
        n = len(b)
        kind = i % 5
        if kind == 0:
            a, e = 0, int(rng.integers(13, n))                       # the front
        elif kind == 1:
            a, e = int(rng.integers(0, n - 20)), n - 1                 # the back
        elif kind == 2:
            a = int(rng.integers(0, n - 13)); e = a + int(rng.integers(0, 12))      # shorter than a word: the whole read
        elif kind == 3:
            a, e = int(rng.integers(0, n - 30)), n + 7                 # beyond the end: clipped
        else:
            a = int(rng.integers(0, n - 14)); e = a + 12                # exactly one word
        ranges.append((a, e))
        prevmax.append((int(rng.integers(0, 60)), int(rng.integers(0, 30))) if i % 2 else (0, 0))
    gp = gix.default_params()
    gp.rmapflg |= api.FLG_NOSHRTINFO | api.FLG_SENSITIVE
    gp.min_basqval = 8
    op = ol.default_params(oix)
    op.flags |= ol.FLG_NOSHRTINFO | ol.FLG_SENSITIVE | ol.FLG_RAWRESULTS
    op.min_basq = 8
    mp = api.Mapper(gix, len(rd), max(len(r[1]) for r in rd))
    mp.set_debug(1)
    om = ol.Mapper(oix)
    nres = 0
    try:
        res, stats, cf = mp.map_batch_ctx([r[1] for r in rd], [r[2] for r in rd], gp, prev_max=prevmax, raw_alignments=True, seed_range=ranges)
        for i, (nm, b, q) in enumerate(rd):
            rv, exp = om.map(b, q, op, prevmax=prevmax[i], seed_range=ranges[i])
            assert rv == 0 and stats[i]["err"] == 0, (i, rv, stats[i]["err"])
            want = pr.stage_lines(om.dump(i, nm))
            got = pr.stage_lines(mp.dump_read(i, nm))
            for x, y in zip(got, want):
                assert x == y, (i, ranges[i], x, y)
            assert len(got) == len(want)
            assert res[i] == exp, (i, ranges[i])
            assert cf[i] == om.cand_first
            st = om.stats()
            assert (stats[i]["swmax"], stats[i]["sw2nd"], stats[i]["nseg"], stats[i]["nseg_tot"], stats[i]["nhit"], stats[i]["nhit_tot"]) == tuple(st[0:6]), i
            nres += len(exp)
    finally:
        mp.close()
        om.close()
        gix.close()
        ol.lib().or_index_free(oix)
    assert nres > 200
