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
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", out, os.path.join(ROOT, "tests", "hostemu", "emu_main.cpp")], check=True)
    return out


@pytest.mark.parametrize("window", [0, 160], ids=["w-default", "w160"])
@pytest.mark.parametrize("entry", gu.MANIFEST, ids=[e["tag"] for e in gu.MANIFEST])
def test_stage_logic_matches_reference_dump(entry, window, emu_bin, oracle_built, tmp_path):
    """window=160: strands with more hits stream through the candidate stage in windows of ascending
    diagonal (and fall back to the HBM working set when a hit region exceeds the window)."""
    fx = gu.unpack(entry, tmp_path)
    env = dict(os.environ, EMU_WINDOW=str(window))
    out = subprocess.run([emu_bin] + entry["opts"].split() + [fx["prefix"], fx["fq"]], check=True, capture_output=True, text=True, env=env).stdout
    a, b = out.split("\n"), fx["expected"].split("\n")
    for i, (x, y) in enumerate(zip(a, b)):
        assert x == y, "line %d" % (i + 1)
    assert len(a) == len(b)
