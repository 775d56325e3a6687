"""Index construction on the GPU (smaltgpu_index_build / _save) against the reference's own index
files: the committed fixtures carry the md5 of the `.sma` / `.smi` files the reference's `smalt index` wrote for them
(tests/golden/make_golden.py), so equality here is byte-for-byte equality with the reference's files.  Plus seeded
references with N runs, lower case, IUPAC letters and sequence lengths off the sampling grid against the oracle's
builder, and mapping through a built image against mapping through the same image loaded from its files."""
import os

import numpy as np
import pytest

import golden_util as gu
import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("entry", gu.MANIFEST, ids=[e["tag"] for e in gu.MANIFEST])
def test_built_index_files_equal_the_reference_files(entry, oracle_built, tmp_path):
    from smalt_amd import api
    fx = gu.unpack(entry, tmp_path)
    ix = api.Index.build(fx["seqs"], fx["names"], entry["k"], entry["s"], 0)
    try:
        pre = str(tmp_path / "gpu_built")
        ix.save(pre)
    finally:
        ix.close()
    assert gu.md5(pre + ".sma") == entry["sma_md5"]
    assert gu.md5(pre + ".smi") == entry["smi_md5"]


def _messy_reference(seed, nseq, lo, hi):
    rng = np.random.default_rng(seed)
    seqs = []
    for _ in range(nseq):
        n = int(rng.integers(lo, hi))
        a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
        for _ in range(int(rng.integers(1, 6))):                      # runs of N
            p = int(rng.integers(0, n - 50)); a[p:p + int(rng.integers(1, 300))] = ord("N")
        for _ in range(20):                                            # isolated IUPAC letters, U for T
            a[int(rng.integers(0, n))] = ord("RYKMSWU"[int(rng.integers(0, 7))])
        low = rng.random(n) < 0.2
        a[low] |= 0x20                                                 # lower case
        seqs.append(a.tobytes())
    return seqs


@pytest.mark.parametrize("k,s", [(13, 6), (11, 2), (20, 13), (9, 6), (17, 1), (8, 8)], ids=lambda v: str(v))
def test_built_index_equals_oracle_builder_on_messy_references(k, s, oracle_built, tmp_path):
    from smalt_amd import api
    seqs = _messy_reference(100 + k * 31 + s, 5, 30_000, 90_001)
    names = ["seq%d with description" % i if i == 2 else "s%d" % i for i in range(len(seqs))]
    oix = ol.build_index(seqs, names, k, s)
    opre = str(tmp_path / "oracle")
    assert ol.lib().or_index_write(oix, opre.encode()) == 0
    ol.lib().or_index_free(oix)
    ix = api.Index.build(seqs, names, k, s, 0)
    try:
        gpre = str(tmp_path / "gpu")
        ix.save(gpre)
        assert ix.build_ms > 0
    finally:
        ix.close()
    for ext in (".sma", ".smi"):
        assert open(opre + ext, "rb").read() == open(gpre + ext, "rb").read(), ext


def test_mapping_through_a_built_image_equals_mapping_through_its_files(oracle_built, tmp_path):
    from smalt_amd import api, synth
    ch = synth.make_reference(4, 300_001, seed=5, repeat_frac=0.1, n_fam=3, cons_len=300, divergence=0.05)
    seqs = [synth.codes_to_ascii(c) for c in ch]
    names = ["c%d" % i for i in range(4)]
    reads, _ = synth.make_reads(ch, 400, 120, seed=6)
    rb = [synth.codes_to_ascii(r) for r in reads]
    built = api.Index.build(seqs, names, 13, 6, 0)
    pre = str(tmp_path / "ix")
    built.save(pre)
    loaded = api.Index.load(pre, 0)
    out = []
    for ix in (built, loaded):
        mp = api.Mapper(ix, len(rb), 120)
        try:
            out.append(mp.map_batch(rb, [b"I" * len(r) for r in rb], ix.default_params()))
        finally:
            mp.close()
    built.close(); loaded.close()
    assert out[0][0] == out[1][0]
    assert out[0][1] == out[1][1]
    assert sum(1 for r in out[0][0] if r) > 300


def test_index_build_rejects_what_the_reference_rejects():
    """A sequence shorter than the word length (hashidx.c:499) and word lengths the index types do not cover."""
    from smalt_amd import api
    with pytest.raises(api.SmaltGpuError):
        api.Index.build([b"ACGTACGTACGTACGTACGTACGT" * 50, b"ACGTACG"], ["a", "b"], 13, 6, 0)
    with pytest.raises(api.SmaltGpuError):
        api.Index.build([b"ACGT" * 5000], ["a"], 22, 6, 0)
    ix = api.Index.build([b"ACGT" * 5000], ["a"], 13, 6, 0)       # and a loaded handle cannot be saved, only a built one
    ix.close()
