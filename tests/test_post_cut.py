"""Cutting an alignment string to a window of reference offsets (smgpost::cut_window, smalt_amd/csrc/smg_post.hpp: the
operation behind the pieces of an alignment that runs across reference sequences, results.c:1472) against 24 000 vectors
the reference's own diffStrSegment produced (tests/golden/make_golden_cut.py): piece string, the four end offsets and the
three outcomes (piece, nothing inside the window, no matched base behind the window's start)."""
import gzip
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cut_window_matches_reference_vectors(tmp_path):
    exe = str(tmp_path / "cut_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "hostemu", "cut_check.cpp")], check=True)
    with gzip.open(os.path.join(ROOT, "tests", "golden", "kat_diffstr_cut.json.gz"), "rt") as g:
        vec = json.load(g)
    assert len(vec) >= 20000
    text = "".join("%s %d %d\n" % (v["s"], v["lo"], v["hi"]) for v in vec)
    out = subprocess.run([exe], input=text, check=True, capture_output=True, text=True).stdout.split("\n")
    seen = {0: 0, 1: 0, -1: 0}
    for i, v in enumerate(vec):
        f = out[i].split()
        code = int(f[0])
        seen[code] += 1
        if v["rv"] == 0:
            assert code == 0, (i, v, out[i])
            assert f[1] == v["out"] and [int(f[2]), int(f[3])] == v["ref"] and [int(f[4]), int(f[5])] == v["read"], (i, v, out[i])
        else:
            # the reference: ERRCODE_NOMATCH (nothing of the alignment inside the window) or a failure code
            assert code == (1 if v["rv"] > 0 else -1), (i, v, out[i])
    assert seen[0] > 20000 and seen[1] > 500 and seen[-1] > 10
