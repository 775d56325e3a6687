"""The CPU oracle against the split-read dumps of the reference itself (tests/golden/gs_*, split_replay.py): every mapSingleRead
call of rmapSingle under RMAPFLG_SPLIT -- the second one with its k-mer words from the stretch the rule of mapSecondary
(rmap.c:1459-1481) names, appended to the set of the first -- reproduces the reference's stage state, alignments and counters.
This pins or_collect_hitinfo's range form (hashhit.c:536-551) and the rule itself to the reference."""
import pytest

import oracle_lib as ol
import pair_replay as pr
import split_replay as sr


@pytest.mark.parametrize("entry", sr.MANIFEST, ids=[e["tag"] for e in sr.MANIFEST])
def test_oracle_replays_both_calls_of_split_reads(entry, oracle_built, tmp_path):
    fx = sr.load_fixture(entry, tmp_path)
    oix = ol.lib().or_index_read(fx["prefix"].encode())
    om = ol.Mapper(oix)
    calls = sr.planned_calls(fx, entry)
    assert len(calls) == entry["calls"] and sum(1 for _, _, r in calls if r) == entry["second_calls"] > 20
    try:
        for no, c, rng in calls:
            nm, b, q = fx["reads"][no]
            op = ol.default_params(oix)
            op.min_swatscor, op.min_cover, op.min_swatscor_below_max, op.min_basq = c["minscor"], c["mincov"], c["belowmax"], fx["min_basq"]
            op.flags = (c["flags"] & (ol.FLG_BEST | ol.FLG_SEQBYSEQ | ol.FLG_NOSHRTINFO | ol.FLG_SENSITIVE)) | ol.FLG_RAWRESULTS
            rv, res = om.map(b, q, op, prevmax=c["prevmax"], seed_range=rng)
            assert rv == c["err"] == 0
            lines = pr.stage_lines(om.dump(no, nm))
            try:
                pr.check_call(c, lines, res, om.cand_first, om.stats())
            except AssertionError as e:
                raise AssertionError("read %d, call with stretch %s: %s" % (no, rng, str(e)[:600]))
    finally:
        om.close()
        ol.lib().or_index_free(oix)


@pytest.fixture(scope="session")
def split_check(tmp_path_factory):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-s", "-C", os.path.join(root, "smalt_amd", "csrc")], check=True)
    exe = str(tmp_path_factory.mktemp("sc") / "split_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-pthread", "-o", exe, os.path.join(root, "tests", "hostemu", "split_check.cpp"),
                    "-L" + os.path.join(root, "smalt_amd"), "-lsmaltgpu", "-Wl,-rpath," + os.path.join(root, "smalt_amd"), "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    return exe


@pytest.mark.parametrize("entry", sr.MANIFEST, ids=[e["tag"] for e in sr.MANIFEST])
def test_split_runner_on_replayed_calls_leaves_the_reference_sets(entry, split_check, oracle_built, tmp_path):
    """the product's split-read runner (smalt_amd/csrc/smg_split.hpp) without a device: tests/hostemu/split_check.cpp feeds it the
    alignments the reference's calls added; it must ask for the second call exactly where the reference made one (with the running
    maxima the reference passed) and leave the reference's sets -- rows, mapping probabilities, sorted and per-segment order"""
    import subprocess
    fx = sr.load_fixture(entry, tmp_path)
    import gzip
    import os
    import golden_util as gu
    dump = str(tmp_path / "dump.txt")
    with gzip.open(os.path.join(gu.GOLD, entry["tag"] + ".refdump.txt.gz"), "rb") as g, open(dump, "wb") as f:
        f.write(g.read())
    seqinfo = str(tmp_path / "seqinfo.txt")
    with open(seqinfo, "w") as o:
        for nm, sq in zip(fx["names"], fx["seqs"]):
            o.write("%s %d\n" % (nm, len(sq)))
    for threads in (1, 3):
        r = subprocess.run([split_check, dump, fx["fq"], seqinfo, "k=%d" % entry["k"], "s=%d" % entry["s"], "threads=%d" % threads], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        got = [ln for ln in r.stdout.split("\n") if ln]
        want = []
        for R in fx["dump"]:
            want += R["post_final"]
        assert len(got) == len(want), (len(got), len(want))
        for i, (x, y) in enumerate(zip(got, want)):
            if x.startswith("RF"):                     # the probability to the last bit: compare as numbers, the rest as text
                fx_, fy = x.split(), y.split()
                assert fx_[:5] == fy[:5] and float(fx_[5]) == float(fy[5]) and fx_[6:] == fy[6:], (i, x, y)
            else:
                assert x == y, (i, x, y)
