"""Larger seeded workloads on the GPU against the CPU oracle (results + per-read scalars of every
read), including repeat-rich reads that exercise the hit-list allocation-boundary protocol, and
the torch-built index image against the oracle's builder."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _oracle_map_all(oix, reads, par):
    m = ol.Mapper(oix)
    out = []
    for r in reads:
        rv, res = m.map(r, b"I" * len(r), par)
        assert rv == 0
        st = m.stats()
        out.append((res, dict(swmax=st[0], sw2nd=st[1], nseg=st[2], nseg_tot=st[3], nhit=st[4], nhit_tot=st[5])))
    m.close()
    return out


@pytest.mark.parametrize("window", [0, 300, -1024], ids=["w-default", "w300", "deferred-cands"])
def test_repeat_rich_genome_matches_oracle(window, oracle_built, tmp_path, monkeypatch):
    """window=300 shrinks the LDS window of the candidate stage so that most strands are streamed in
    several windows and a few fall back to the HBM working set (a hit region larger than the window).
    -1024: first-pass candidate slots of 1024 hits per strand -- every repeat read overflows its slot and is deferred to the
    second pass over the few worst-case slots (the path a read with more hits than 4 x the reference's hit-list allocation
    takes: 1 in 2 M reads of the bench's reference)."""
    from smalt_amd import api, synth
    if window > 0:
        monkeypatch.setenv("SMALTGPU_CANDS_WINDOW", str(window))
    if window < 0:
        monkeypatch.setenv("SMALTGPU_CANDS_HCAP", str(-window))
    ch = synth.make_reference(4, 2_500_000, seed=21, repeat_frac=0.15, n_fam=1, cons_len=300, divergence=0.05)
    reads, _ = synth.make_reads(ch, 1500, 100, seed=22, sub_rate=0.01, indel_read_frac=0.05)
    seqs = [synth.codes_to_ascii(c) for c in ch]
    rb = [synth.codes_to_ascii(r) for r in reads]
    oix0 = ol.build_index(seqs, ["c%d" % i for i in range(4)], 13, 6)
    pre = str(tmp_path / "rep")
    assert ol.lib().or_index_write(oix0, pre.encode()) == 0
    oix = ol.lib().or_index_read(pre.encode())
    exp = _oracle_map_all(oix, rb, ol.default_params(oix))
    # with 100 bp reads the hit list holds 16384 words (hashhit.c:1266): some repeat reads exceed it
    # and take the allocation-boundary retry protocol (checked on the CPU: 3 of the first 600 reads)
    gix = api.Index.load(pre, 0)
    mp = api.Mapper(gix, len(rb), 100)
    try:
        res, stats = mp.map_batch(rb, [b"I" * len(r) for r in rb], gix.default_params())
    finally:
        mp.close()
        gix.close()
    for i in range(len(rb)):
        assert stats[i]["err"] == 0
        assert res[i] == exp[i][0], i
        for kk, v in exp[i][1].items():
            assert stats[i][kk] == v, (i, kk)


def test_mid_length_reads_take_the_long_window_kernels(oracle_built, tmp_path):
    """220 bp reads: every reference window exceeds the small-LDS instance of the packed K2a kernel, so the ranked
    candidates go through the listed large-window instance; a tenth of the reads carry an N (32-bit kernel)."""
    from smalt_amd import api, synth
    ch = synth.make_reference(3, 400_000, seed=31, repeat_frac=0.1, n_fam=3, cons_len=300, divergence=0.05)
    reads, _ = synth.make_reads(ch, 300, 220, seed=32, sub_rate=0.02, indel_read_frac=0.3)
    seqs = [synth.codes_to_ascii(c) for c in ch]
    rb = []
    rng = np.random.default_rng(33)
    for i, r in enumerate(reads):
        a = bytearray(synth.codes_to_ascii(r))
        if i % 10 == 0:
            a[int(rng.integers(0, len(a)))] = ord("N")
        rb.append(bytes(a))
    oix0 = ol.build_index(seqs, ["c%d" % i for i in range(3)], 13, 6)
    pre = str(tmp_path / "mid")
    assert ol.lib().or_index_write(oix0, pre.encode()) == 0
    oix = ol.lib().or_index_read(pre.encode())
    exp = _oracle_map_all(oix, rb, ol.default_params(oix))
    gix = api.Index.load(pre, 0)
    mp = api.Mapper(gix, len(rb), 220)
    try:
        res, stats = mp.map_batch(rb, [b"I" * len(r) for r in rb], gix.default_params())
    finally:
        mp.close()
        gix.close()
    for i in range(len(rb)):
        assert stats[i]["err"] == 0
        assert res[i] == exp[i][0], i
        for kk, v in exp[i][1].items():
            assert stats[i][kk] == v, (i, kk)


def test_torch_index_equals_oracle_index(oracle_built, tmp_path):
    import torch
    from smalt_amd import api, gpuindex, synth
    dev = torch.device("cuda", 0)
    nchr, chrlen, k, s = 3, 300_001, 9, 6
    ref = gpuindex.make_reference_gpu(nchr, chrlen, 5, dev, repeat_frac=0.1, n_fam=3)
    sop = np.arange(nchr + 1, dtype=np.int64) * chrlen
    packed = gpuindex.pack_reference(ref)
    idx, pos = gpuindex.build_perfect_index(ref, sop, k, s)
    codes = ref.cpu().numpy()
    seqs = [synth.codes_to_ascii(codes[i * chrlen:(i + 1) * chrlen]) for i in range(nchr)]
    oix = ol.build_index(seqs, ["c%d" % i for i in range(nchr)], k, s)
    o = oix.contents
    assert o.typ == 0
    assert np.array_equal(np.ctypeslib.as_array(o.idx, shape=(o.nkeys + 1,)), idx.cpu().numpy().view(np.uint32))
    assert np.array_equal(np.ctypeslib.as_array(o.pos, shape=(o.npos,)), pos.cpu().numpy().view(np.uint32))
    assert np.array_equal(np.ctypeslib.as_array(o.packed, shape=(o.totlen // 10 + 1,)), packed.cpu().numpy().view(np.uint32))
    # map through the adopted device image
    reads_ascii, _ = gpuindex.make_reads_gpu(ref, sop, 400, 100, 9)
    rb = [bytes(x) for x in reads_ascii.cpu().numpy().reshape(400, 100)]
    desc = api.IndexDesc()
    desc.k, desc.s, desc.typ, desc.nbits_key, desc.nbits_lo = k, s, 0, 2 * k, 0
    desc.npos, desc.nwords = int(pos.numel()), 0
    desc.idx, desc.pos, desc.packed = idx.data_ptr(), pos.data_ptr(), packed.data_ptr()
    desc.nseq = nchr
    sop_u64 = np.ascontiguousarray(sop.astype(np.uint64))
    desc.sop = sop_u64.ctypes.data
    desc.on_device = 1
    torch.cuda.synchronize()
    gix = api.Index.from_desc(desc, 0)
    mp = api.Mapper(gix, 400, 100)
    try:
        res, stats = mp.map_batch(rb, None, gix.default_params())
    finally:
        mp.close()
        gix.close()
    exp = _oracle_map_all(oix, rb, ol.default_params(oix))
    for i in range(len(rb)):
        assert res[i] == exp[i][0], i


def test_candidate_pool_overflow_is_recovered(oracle_built, tmp_path):
    """The mapper's pools are sized for the average read; the reference's buffers grow.  With a pool of 8 ranked candidates
    per read (instead of hundreds) nearly every batch overflows: smaltgpu_map_batch must re-map the reads that did not fit
    in smaller batches and return exactly what the oracle returns -- never an error."""
    from smalt_amd import api, synth
    ch = synth.make_reference(2, 600_000, seed=41, repeat_frac=0.3, n_fam=1, cons_len=300, divergence=0.03)
    reads, _ = synth.make_reads(ch, 1200, 100, seed=42, sub_rate=0.01, indel_read_frac=0.05)
    seqs = [synth.codes_to_ascii(c) for c in ch]
    rb = [synth.codes_to_ascii(r) for r in reads]
    oix0 = ol.build_index(seqs, ["c0", "c1"], 11, 3)
    pre = str(tmp_path / "ovf")
    assert ol.lib().or_index_write(oix0, pre.encode()) == 0
    oix = ol.lib().or_index_read(pre.encode())
    exp = _oracle_map_all(oix, rb, ol.default_params(oix))
    assert max(e[1]["nseg"] for e in exp) > 64          # reads that rank far more candidates than the pool gives them
    gix = api.Index.load(pre, 0)
    try:
        for per in (8, 40):
            mp = api.Mapper(gix, len(rb), 100, cands_per_read=per)
            try:
                for rep in range(2):                        # the mapper stays usable after a recovery
                    res, stats = mp.map_batch(rb, [b"I" * len(r) for r in rb], gix.default_params())
                    for i in range(len(rb)):
                        assert stats[i]["err"] == 0
                        assert res[i] == exp[i][0], (per, i)
                        for kk, v in exp[i][1].items():
                            assert stats[i][kk] == v, (per, i, kk)
            finally:
                mp.close()
    finally:
        gix.close()


def test_second_index_image_is_a_device_copy(oracle_built, tmp_path):
    """smaltgpu_index_clone (device-to-device copy; on a one-GPU box onto the same device) gives an image that maps like the
    original, and survives the original being freed."""
    from smalt_amd import api, synth
    ch = synth.make_reference(2, 200_000, seed=43, repeat_frac=0.1, n_fam=2, cons_len=300)
    reads, _ = synth.make_reads(ch, 200, 100, seed=44)
    rb = [synth.codes_to_ascii(r) for r in reads]
    for k, s in ((13, 6), (20, 13)):                     # hash32mix without and with perfect bits
        g1 = api.Index.build([synth.codes_to_ascii(c) for c in ch], ["c0", "c1"], k, s, 0)
        g2 = g1.clone(api.device_count() - 1)
        m1 = api.Mapper(g1, len(rb), 100)
        r1 = m1.map_batch(rb, None, g1.default_params())
        m1.close()
        g1.close()
        m2 = api.Mapper(g2, len(rb), 100)
        r2 = m2.map_batch(rb, None, g2.default_params())
        m2.close()
        g2.close()
        assert r1 == r2 and sum(1 for x in r1[0] if x) > 150


@pytest.mark.parametrize("k,s,nreads,dircap", [(20, 13, 14, 0), (13, 6, 40, 0), (13, 6, 40, 65536), (13, 6, 40, -4096)],
                         ids=["k20s13", "k13s6", "k13s6-deferred-k3", "k13s6-deferred-cands"])
def test_long_reads_match_oracle(k, s, nreads, dircap, oracle_built, tmp_path, monkeypatch):
    """BASELINE configs[4] shape in small: reads of 0.3-3 kbp with 3/5/4 % substitutions/insertions/deletions (every
    fifth read an exact copy: narrow bands, K2b), k=20 s=13 (HASH32MIX with nbits_perf) and k=13 s=6 (many more hits):
    wave-parallel candidate stage for long reads, strip K2a kernel, wide-band K3 in two passes (small direction-matrix slots
    for all reads, full-size slots for the reads a band of which does not fit)."""
    from smalt_amd import api, synth
    if dircap > 0:  # first-pass direction matrices of 64 KB: most reads are deferred to the second K3 pass (full-size slots)
        monkeypatch.setenv("SMALTGPU_ALIGN_DIRCAP", str(dircap))
    if dircap < 0:  # first-pass candidate slots of 4096 hits per strand: most reads are deferred to the second candidate pass
        monkeypatch.setenv("SMALTGPU_CANDS_HCAP", str(-dircap))
    ch = synth.make_reference(3, 700_000, seed=51, repeat_frac=0.1, n_fam=3, cons_len=400, divergence=0.05)
    rng = np.random.default_rng(52)
    seqs = [synth.codes_to_ascii(c) for c in ch]
    rb = []
    for i in range(nreads):
        c = int(rng.integers(0, 3))
        ln = 1500 if nreads == 14 else int(rng.integers(300, 3000))
        p = int(rng.integers(0, 700_000 - ln - 200))
        src = bytearray(seqs[c][p:p + ln])
        out = bytearray()
        for ch_ in src:
            u = rng.random()
            if nreads != 14 and i % 5 == 0:
                out.append(ch_)
            elif u < 0.03:
                out.append(b"ACGT"[int(rng.integers(0, 4))])
            elif u < 0.08:
                out.append(ch_); out.append(b"ACGT"[int(rng.integers(0, 4))])
            elif u < 0.12:
                continue
            else:
                out.append(ch_)
        r = bytes(out)
        if i % 2:
            r = r[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
        rb.append(r)
    oix0 = ol.build_index(seqs, ["c%d" % i for i in range(3)], k, s)
    pre = str(tmp_path / "long")
    assert ol.lib().or_index_write(oix0, pre.encode()) == 0
    oix = ol.lib().or_index_read(pre.encode())
    exp = _oracle_map_all(oix, rb, ol.default_params(oix))
    gix = api.Index.load(pre, 0)
    mp = api.Mapper(gix, len(rb), max(len(r) for r in rb))
    try:
        res, stats = mp.map_batch(rb, [b"5" * len(r) for r in rb], gix.default_params())
        ms, _ = mp.timers()
        print("long-read kernel ms:", {k_: round(v, 2) for k_, v in ms.items()})
    finally:
        mp.close()
        gix.close()
    for i in range(len(rb)):
        assert stats[i]["err"] == 0
        assert res[i] == exp[i][0], i
        for kk, v in exp[i][1].items():
            assert stats[i][kk] == v, (i, kk)


def test_edge_batches(oracle_built, tmp_path):
    """Empty batch, reads shorter than k, a read of exactly the mapper's maximum length, an over-long read, all-N reads."""
    from smalt_amd import api, synth
    ch = synth.make_reference(2, 150_000, seed=61, repeat_frac=0.05, n_fam=1, cons_len=200)
    seqs = [synth.codes_to_ascii(c) for c in ch]
    oix0 = ol.build_index(seqs, ["c0", "c1"], 13, 6)
    pre = str(tmp_path / "edge")
    assert ol.lib().or_index_write(oix0, pre.encode()) == 0
    oix = ol.lib().or_index_read(pre.encode())
    gix = api.Index.load(pre, 0)
    mp = api.Mapper(gix, 8, 120)
    par = gix.default_params()
    try:
        res, stats = mp.map_batch([], None, par)
        assert res == [] and stats == []
        rb = [b"ACGTA", b"N" * 40, seqs[0][1000:1120], seqs[1][500:512], b"A"]
        res, stats = mp.map_batch(rb, None, par)
        exp = _oracle_map_all(oix, rb, ol.default_params(oix))
        for i in range(len(rb)):
            assert stats[i]["err"] == 0
            assert res[i] == exp[i][0], i
        assert res[0] == [] and res[1] == [] and len(res[2]) >= 1
        with pytest.raises(api.SmaltGpuError):
            mp.map_batch([seqs[0][:121]], None, par)               # longer than the mapper was created for
        with pytest.raises(api.SmaltGpuError):
            mp.map_batch([b"ACGT"] * 9, None, par)                 # more reads than the mapper was created for
        res, stats = mp.map_batch([seqs[0][2000:2100]], None, par)  # still usable afterwards
        assert len(res[0]) >= 1
    finally:
        mp.close()
        gix.close()


def test_8kbp_reads_match_oracle(oracle_built, tmp_path):
    """BASELINE configs[4] at its real read length: 8 kbp sources with 3/5/4 % substitutions/insertions/deletions
    (PacBio shape), k=20 s=13, both strands, one exact copy (narrow bands: K2b) -- the 8-strip K2a kernel, K3 strips of
    >= 1024 columns and the two-pass candidate / direction-matrix slots, against the oracle read by read."""
    from smalt_amd import api, synth
    ch = synth.make_reference(3, 700_000, seed=71, repeat_frac=0.1, n_fam=3, cons_len=400, divergence=0.05)
    seqs = [synth.codes_to_ascii(c) for c in ch]
    reads, _ = synth.make_long_reads(ch, 7, 8000, seed=72, sub=0.03, ins=0.05, dele=0.04)
    rb = [synth.codes_to_ascii(r) for r in reads]
    rb.append(seqs[1][300_000:308_000])                    # error-free 8 kbp read
    assert min(len(r) for r in rb) > 7800
    oix0 = ol.build_index(seqs, ["c%d" % i for i in range(3)], 20, 13)
    pre = str(tmp_path / "l8k")
    assert ol.lib().or_index_write(oix0, pre.encode()) == 0
    oix = ol.lib().or_index_read(pre.encode())
    exp = _oracle_map_all(oix, rb, ol.default_params(oix))
    gix = api.Index.load(pre, 0)
    mp = api.Mapper(gix, len(rb), max(len(r) for r in rb))
    try:
        res, stats = mp.map_batch(rb, [b"5" * len(r) for r in rb], gix.default_params())
        ms, _ = mp.timers()
        print("8 kbp kernel ms:", {k_: round(v, 2) for k_, v in ms.items()})
    finally:
        mp.close()
        gix.close()
    for i in range(len(rb)):
        assert stats[i]["err"] == 0
        assert res[i], i
        assert res[i] == exp[i][0], i
        for kk, v in exp[i][1].items():
            assert stats[i][kk] == v, (i, kk)
