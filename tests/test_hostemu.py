"""The per-read stage logic the HIP kernels execute (smalt_amd/csrc/smg_stages.hpp), compiled for
the host with one lane (tests/hostemu), against the committed reference dumps.  This checks the
product's own sequential logic on a machine without a GPU; the GPU tests check the kernels."""
import os
import subprocess

import pytest

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def emu_bin():
    out = os.path.join(ROOT, "tests", "hostemu", "emu")
    tmp = "%s.%d" % (out, os.getpid())             # pytest-xdist workers build at the same time: never run a half-written file
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", tmp, os.path.join(ROOT, "tests", "hostemu", "emu_main.cpp")], check=True)
    os.replace(tmp, out)
    return out


@pytest.mark.parametrize("window,split", [(0, 0), (160, 0), (0, 1), (160, 1)], ids=["w-default", "w160", "split", "split-w160"])
@pytest.mark.parametrize("entry", gu.MANIFEST, ids=[e["tag"] for e in gu.MANIFEST])
def test_stage_logic_matches_reference_dump(entry, window, split, emu_bin, oracle_built, tmp_path):
    """window=160: strands with more hits stream through the candidate stage in windows of ascending
    diagonal (and fall back to the HBM working set when a hit region exceeds the window).
    split: S3 as a stage of its own (stage_hits = kernel k_hits: gather + sort into the batch-wide pool, here with windows of
    320 keys), the candidate stage streaming the sorted keys in chunks of `window` (or the LDS working set)."""
    fx = gu.unpack(entry, tmp_path)
    env = dict(os.environ, EMU_WINDOW=str(window), EMU_SPLIT=str(split), EMU_HITS_WINDOW="400")
    out = subprocess.run([emu_bin] + entry["opts"].split() + [fx["prefix"], fx["fq"]], check=True, capture_output=True, text=True, env=env).stdout
    a, b = out.split("\n"), fx["expected"].split("\n")
    for i, (x, y) in enumerate(zip(a, b)):
        assert x == y, "line %d" % (i + 1)
    assert len(a) == len(b)


@pytest.mark.parametrize("history", [1, 0], ids=["serial-order", "stateless"])
@pytest.mark.parametrize("split", [0, 1], ids=["fused", "split"])
def test_hit_list_capacity_follows_the_serial_run(history, split, emu_bin, oracle_built, tmp_path):
    """tests/golden/make_golden_history.py: 100-base reads on a 430-copy repeat reach the allocation boundary of the reference's hit
    list (hashhit.c:1497, :1730-1741) unless a longer read has grown the list before them (hashhit.c:1280-1282).  With the lengths
    of the longest earlier read per read (Batch::alloc_len, EMU_HISTORY=1) the stage logic gives the dump of the reference run
    serially over the file; without, the dump of the reference run on each read alone."""
    import gzip
    import json
    entry = json.load(open(os.path.join(gu.GOLD, "manifest_history.json")))
    fx = gu.unpack(entry, tmp_path)
    expected = fx["expected"] if history else gzip.open(os.path.join(gu.GOLD, entry["tag"] + ".refdump_fresh.txt.gz"), "rt").read()
    assert fx["expected"] != expected or history
    env = dict(os.environ, EMU_HISTORY=str(history), EMU_SPLIT=str(split), EMU_HITS_WINDOW="400")
    out = subprocess.run([emu_bin] + entry["opts"].split() + [fx["prefix"], fx["fq"]], check=True, capture_output=True, text=True, env=env).stdout
    a, b = out.split("\n"), expected.split("\n")
    for i, (x, y) in enumerate(zip(a, b)):
        assert x == y, "line %d" % (i + 1)
    assert len(a) == len(b)


def _mutate(rng, src, sub, ins, dele):
    out = bytearray()
    for ch in src:
        u = rng.random()
        if u < sub:
            out.append(b"ACGT"[int(rng.integers(0, 4))])
        elif u < sub + ins:
            out.append(ch)
            out.append(b"ACGT"[int(rng.integers(0, 4))])
        elif u < sub + ins + dele:
            continue
        else:
            out.append(ch)
    return bytes(out)


@pytest.mark.parametrize("k,s,opts", [(13, 6, ""), (20, 13, ""), (13, 6, "-x")], ids=["k13s6", "k20s13", "k13s6-x"])
def test_long_reads_wave_form_equals_sequential_form(k, s, opts, emu_bin, oracle_built, tmp_path):
    """Reads of 256 bases and more: the wave-parallel candidate stage (wide covers, coverage masks in memory, 64-bit
    ranking words) against the sequential restatement, which the committed reference dumps pin.  The reads carry
    substitutions and indels (hit regions of many segments), some are exact copies (narrow bands: K2b)."""
    import numpy as np
    import oracle_lib as ol
    from smalt_amd import synth
    ch = synth.make_reference(3, 400_000, seed=71, repeat_frac=0.2, n_fam=4, cons_len=500, divergence=0.04)
    seqs = [synth.codes_to_ascii(c) for c in ch]
    rng = np.random.default_rng(72)
    fq = tmp_path / "long.fq"
    with open(fq, "wb") as f:
        for i in range(24):
            c = int(rng.integers(0, 3))
            ln = int(rng.integers(256, 3000))
            p = int(rng.integers(0, 400_000 - ln - 1))
            src = seqs[c][p:p + ln]
            r = src if i % 6 == 0 else _mutate(rng, src, 0.02, 0.03, 0.03)
            if i % 2:
                r = r[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))
    oix = ol.build_index(seqs, ["c%d" % i for i in range(3)], k, s)
    pre = str(tmp_path / "longix")
    assert ol.lib().or_index_write(oix, pre.encode()) == 0
    ol.lib().or_index_free(oix)
    outs = []
    for force in ("v1", ""):
        env = dict(os.environ, EMU_CANDS=force)
        outs.append(subprocess.run([emu_bin] + opts.split() + [pre, str(fq)], check=True, capture_output=True, text=True, env=env).stdout)
    a, b = outs[0].split("\n"), outs[1].split("\n")
    assert len(a) > 100
    for i, (x, y) in enumerate(zip(a, b)):
        assert x == y, "line %d" % (i + 1)
    assert len(a) == len(b)


def test_branch_free_cell_update_equals_case_tree(tmp_path):
    """K2b / K3 cell update: the product's branch-free form against the reference's case tree, exhaustively over all
    orderings of its operands (tests/hostemu/cell_equiv.cpp)."""
    exe = str(tmp_path / "cell_equiv")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "hostemu", "cell_equiv.cpp")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok ")
