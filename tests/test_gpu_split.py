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
SMALT_GPU = os.path.join(ROOT, "oracle", "_ref", "smalt_gpu")     # the reference program with its mapping worker bound to the library (integration/)


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
    # the reference's own program bound to the library: blocks of reads through smaltgpu_map_split (integration/rmap_gpu.c), its own report
    if os.path.exists(SMALT_GPU):
        env = dict(os.environ, SMALTGPU_INDEX_PREFIX=pre)
        for extra, more in (([], {}), (["-n", "3", "-O"], {"SMALTGPU_BLOCK_READS": "256"})):
            r = subprocess.run([SMALT_GPU, "map", "-p"] + opts + extra + ["-o", out_gpu, pre, fq], capture_output=True, env=dict(env, **more))
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            b = [ln for ln in open(out_gpu).read().split("\n") if not ln.startswith("@PG")]
            diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
            assert not diff and len(a) == len(b), ("bound program", extra, len(a), len(b), diff[:3])
        # ... and it does go through the library: without the index prefix it cannot map
        env.pop("SMALTGPU_INDEX_PREFIX")
        assert subprocess.run([SMALT_GPU, "map", "-p"] + opts + ["-o", out_gpu, pre, fq], capture_output=True, env=env).returncode != 0
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


import split_replay as sr  # noqa: E402


@pytest.mark.parametrize("entry", sr.MANIFEST, ids=[e["tag"] for e in sr.MANIFEST])
def test_gpu_replays_both_calls_of_split_reads(entry, oracle_built, tmp_path):
    """the committed dumps of the reference's own rmapSingle under RMAPFLG_SPLIT (tests/golden/gs_*): every first call in one
    device batch, every second call (k-mer words from the stretch mapSecondary's rule names) in another; stage state line by
    line, alignments appended as results.c:1906-1935 does, the set's running maxima"""
    from smalt_amd import api
    import pair_replay as pr
    fx = sr.load_fixture(entry, tmp_path)
    gix = api.Index.load(fx["prefix"], 0)
    calls = sr.planned_calls(fx, entry)
    maxlen = max(len(r[1]) for r in fx["reads"])
    ndone = 0
    try:
        for second in (False, True):
            sub = [(no, c, rng) for no, c, rng in calls if (rng is not None) == second]
            assert len(sub) > 20
            par = gix.default_params()
            c0 = sub[0][1]
            assert all((c["mincov"], c["flags"], c["belowmax"], c["minscor"]) == (c0["mincov"], c0["flags"], c0["belowmax"], c0["minscor"]) for _, c, _ in sub)
            par.min_cover, par.min_swatscor_below_max, par.min_basqval, par.min_swatscor = c0["mincov"], c0["belowmax"], fx["min_basq"], c0["minscor"]
            par.rmapflg = c0["flags"] & (api.FLG_BEST | api.FLG_SEQBYSEQ | api.FLG_NOSHRTINFO | api.FLG_SENSITIVE)
            rd = [fx["reads"][no] for no, _, _ in sub]
            mp = api.Mapper(gix, len(sub), maxlen)
            mp.set_debug(1)
            try:
                res, stats, cf = mp.map_batch_ctx([r[1] for r in rd], [r[2] for r in rd], par, prev_max=[c["prevmax"] for _, c, _ in sub], raw_alignments=True,
                                                  seed_range=[rng for _, _, rng in sub] if second else None)
                for i, (no, c, rng) in enumerate(sub):
                    st = stats[i]
                    assert st["err"] == 0
                    lines = pr.stage_lines(mp.dump_read(i, rd[i][0]))
                    lines[0] = lines[0].replace("READ %d " % i, "READ %d " % no, 1)
                    try:
                        pr.check_call(c, lines, res[i], cf[i], (st["swmax"], st["sw2nd"], st["nseg"], st["nseg_tot"], st["nhit"], st["nhit_tot"]))
                    except AssertionError as e:
                        raise AssertionError("read %d, stretch %s: %s" % (no, rng, str(e)[:800]))
                    ndone += 1
            finally:
                mp.close()
    finally:
        gix.close()
    assert ndone == entry["calls"]


@pytest.mark.parametrize("entry", sr.MANIFEST, ids=[e["tag"] for e in sr.MANIFEST])
def test_map_split_leaves_the_set_the_reference_leaves(entry, oracle_built, tmp_path):
    """smaltgpu_map_split over the fixture's reads: the alignment set of every read as rmapSingle returns it (behind the `PE`
    line of the dump) -- rows with status, mapping quality, probability to the last bit, sequence, segment and rank, the
    sorted and the per-segment order -- and the number of second calls"""
    from smalt_amd import api
    fx = sr.load_fixture(entry, tmp_path)
    L = api.lib()
    gix = api.Index.load(fx["prefix"], 0)
    MASK = ~(0x10 | 0x20 | 0x200)
    post = L.smaltgpu_post_create()
    par = gix.default_params()
    par.rmapflg |= api.FLG_NOSHRTINFO | api.FLG_SENSITIVE
    par.min_basqval = fx["min_basq"]
    mp = api.Mapper(gix, len(fx["reads"]), max(len(r[1]) for r in fx["reads"]))
    try:
        for nthreads in (1, 3):
            out, nsec = mp.map_split([r[1] for r in fx["reads"]], [r[2] for r in fx["reads"]], par, post, nthreads)
            assert nsec == entry["second_calls"]
            by_no = {R["no"]: R for R in fx["dump"]}
            for i in range(len(fx["reads"])):
                a, e = out.res_off[i], out.res_off[i + 1]
                if i not in by_no or not by_no[i]["post_final"]:
                    assert a == e
                    continue
                st = sr.post_state(by_no[i]["post_final"])
                assert st["ps"][0] == e - a and st["ps"][2] == out.qsegno[i] and st["ps"][3] == out.setstatus[i], (i, st["ps"], e - a, out.qsegno[i], out.setstatus[i])
                for k_, w in enumerate(st["rows"]):
                    r = out.res[a + k_]
                    got = (r.status & MASK, r.swatscor, r.mapscor, repr(r.prob), r.q_start, r.q_end, r.s_start, r.s_end, r.sidx, r.rsltx, r.qsegx, r.swrank,
                           bytes(out.diffstr[r.stroffs:r.stroffs + r.strlen]))
                    exp = (w["status"] & MASK, w["score"], w["mapscor"], repr(w["prob"]), w["q_start"], w["q_end"], w["s_start"], w["s_end"], w["sidx"], w["rsltx"], w["qsegx"],
                           w["swrank"], w["diffstr"])
                    assert got == exp, (i, k_, got, exp)
                assert [out.sortr[x] for x in range(out.sort_off[i], out.sort_off[i + 1])] == st["so"], i
                if st["ss"] is not None:
                    assert [out.segsrtr[x] for x in range(out.sort_off[i], out.sort_off[i + 1])] == st["ss"], i
                    assert [out.segnor[x] for x in range(out.seg_off[i], out.seg_off[i + 1])] == st["sg"], i
    finally:
        mp.close()
        L.smaltgpu_post_free(post)
        gix.close()


def chimeric_pairs(tmp, nchr, chrlen, npairs, rlen, seed):
    """read pairs (fragments of 300 +- 30 bases) of which two in five carry a mate whose tail (or head) comes from somewhere else"""
    from smalt_amd import synth
    ch = synth.make_reference(nchr, chrlen, seed=seed, repeat_frac=0.2, n_fam=2, cons_len=400, divergence=0.03)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ch)
    r1, r2, _ = synth.make_pairs(ch, npairs, rlen, seed=seed + 1, insert_mean=300, insert_sd=30, sub_rate=0.02, indel_read_frac=0.2)
    rng = np.random.default_rng(seed + 2)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    fqs = []
    for which, reads in ((1, r1), (2, r2)):
        fq = os.path.join(tmp, "reads_%d.fq" % which)
        with open(fq, "wb") as f:
            for i, r in enumerate(reads):
                b = bytearray(synth.codes_to_ascii(r))
                if (i + which) % 5 < 2 and len(b) > 80:                 # a stretch from elsewhere, either strand
                    cut = int(rng.integers(30, len(b) - 30))
                    c = int(rng.integers(0, len(ch)))
                    p = int(rng.integers(0, len(ch[c]) - len(b)))
                    alien = synth.codes_to_ascii(ch[c][p:p + len(b) - cut])
                    if rng.random() < 0.5:
                        alien = alien.translate(comp)[::-1]
                    b = b[:cut] + bytearray(alien) if rng.random() < 0.5 else bytearray(alien) + b[len(b) - cut:]
                u = rng.random()
                if u < 0.03:
                    b = b[:int(rng.integers(5, 16))]                    # around the word length
                elif u < 0.06:
                    b = bytearray(synth.codes_to_ascii(rng.integers(0, 4, size=len(b), dtype=np.uint8)))
                if rng.random() < 0.05 and len(b) > 4:
                    b[int(rng.integers(0, len(b)))] = ord("N")
                q = bytes(33 + int(x) for x in rng.integers(5, 41, size=len(b)))
                f.write(b"@p%d/%d\n" % (i, which) + bytes(b) + b"\n+\n" + q + b"\n")
        fqs.append(fq)
    return fa, fqs


@pytest.mark.skipif(not os.path.exists(SMALT), reason="reference binary not built (make -C oracle ref)")
@pytest.mark.parametrize("k,s,nchr,chrlen,rlen,opts", [
    (13, 6, 3, 300_000, 150, ["-f", "cigar", "-i", "500"]),
    (11, 3, 2, 200_000, 120, ["-f", "sam", "-i", "600", "-j", "50", "-q", "10"]),
    (13, 4, 3, 200_000, 200, ["-f", "ssaha", "-i", "500", "-r", "-1"]),
    (13, 6, 600, 2_000, 100, ["-f", "sam:nohead", "-i", "500"]),                 # concatenated mode
    (13, 6, 3, 300_000, 150, ["-f", "cigar", "-i", "500", "-l", "mp", "-m", "30"]),
])
def test_smaltgpu_map_prints_what_smalt_map_prints_for_split_read_pairs(k, s, nchr, chrlen, rlen, opts, tmp_path):
    """rmapPair with RMAPFLG_SPLIT (rmap.c:2073-2097): after its rounds a second call for the read and for the mate, then the pairing;
    the partial alignments of both mates follow the pair's lines (resultpairs.c:1293-1311)"""
    tmp = str(tmp_path)
    if "-r" not in opts:
        opts = opts + ["-r", "3"]
    fa, fqs = chimeric_pairs(tmp, nchr, chrlen, 1200, rlen, seed=k * 1000 + s * 10 + nchr + 5)
    pre = os.path.join(tmp, "idx")
    subprocess.run([SMALT, "index", "-k", str(k), "-s", str(s), pre, fa], check=True, capture_output=True)
    out_ref, out_gpu = os.path.join(tmp, "ref.out"), os.path.join(tmp, "gpu.out")
    subprocess.run([SMALT, "map", "-p"] + opts + ["-o", out_ref, pre] + fqs, check=True, capture_output=True)
    a = [ln for ln in open(out_ref).read().split("\n") if not ln.startswith("@PG")]
    partial = sum(1 for ln in a if ln.startswith("cigar:P") or ln.startswith("alignment:P") or (len(ln.split("\t")) > 1 and ln.split("\t")[1].isdigit() and int(ln.split("\t")[1]) & 0x100))
    assert partial >= 200, partial
    for extra in (["-B", "300"], ["-B", "4000", "-n", "2"]):
        r = subprocess.run([PROG, "-p"] + opts + extra + ["-o", out_gpu, pre] + fqs, capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        b = [ln for ln in open(out_gpu).read().split("\n") if not ln.startswith("@PG")]
        diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
        assert not diff and len(a) == len(b), (len(a), len(b), diff[:3])
    # the reference's own program bound to the library: the second calls are one more round of rmapGpuPairBatch (integration/rmap_gpu.c)
    if os.path.exists(SMALT_GPU):
        env = dict(os.environ, SMALTGPU_INDEX_PREFIX=pre)
        for extra, more in (([], {}), (["-n", "3", "-O"], {"SMALTGPU_BLOCK_READS": "256"}), ([], {"SMALTGPU_NO_COMBINE": "1"})):
            if extra and opts[opts.index("-r") + 1] != "-1":
                continue                          # worker threads share one random generator: only without draws
            r = subprocess.run([SMALT_GPU, "map", "-p"] + opts + extra + ["-o", out_gpu, pre] + fqs, capture_output=True, env=dict(env, **more))
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            b = [ln for ln in open(out_gpu).read().split("\n") if not ln.startswith("@PG")]
            diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
            assert not diff and len(a) == len(b), ("bound program", extra, more, len(a), len(b), diff[:3])
        env.pop("SMALTGPU_INDEX_PREFIX")           # ... through the library: without the index prefix the bound program cannot map
        assert subprocess.run([SMALT_GPU, "map", "-p"] + opts + ["-o", out_gpu, pre] + fqs, capture_output=True, env=env).returncode != 0
