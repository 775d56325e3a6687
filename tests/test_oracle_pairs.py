"""CPU oracle vs the committed PAIRED fixtures (tests/golden/make_golden_pairs.py: dumps of the reference's own rmapPair,
rmap.c:1744): every mapSingleRead call of every pair -- the rare mate, the mate restricted to the intervals the first one
implies (rmap.c:354-492), the unrestricted re-map, the re-map over the on-the-fly k=5 s=1 index (rmap.c:495-517,
:2010-2039) -- is replayed with the call's recorded arguments and must reproduce the reference's stage state line by line,
its alignments, and the result set's running score maxima."""
import json
import os

import pytest

import golden_util as gu
import oracle_lib as ol
import pair_replay as pr

MANIFEST = json.load(open(os.path.join(gu.GOLD, "manifest_pairs.json")))


@pytest.mark.parametrize("entry", MANIFEST, ids=[e["tag"] for e in MANIFEST])
def test_oracle_replays_every_call_of_rmappair(entry, oracle_built, tmp_path):
    fx = pr.load_fixture(entry, tmp_path)
    ix = ol.lib().or_index_read(fx["prefix"].encode())
    m = ol.Mapper(ix)
    ncalls = nfine = nrestr = 0
    try:
        for P in fx["pairs"]:
            for c in P["calls"]:
                name, seq, q = (fx["reads2"] if c["mate"] else fx["reads1"])[P["no"]]
                p = ol.default_params(ix)
                p.min_swatscor, p.min_cover, p.min_swatscor_below_max, p.min_basq = c["minscor"], c["mincov"], c["belowmax"], fx["min_basq"]
                p.flags = c["flags"] & (ol.FLG_BEST | ol.FLG_SEQBYSEQ | ol.FLG_NOSHRTINFO | ol.FLG_SENSITIVE)
                p.flags |= ol.FLG_RAWRESULTS             # every alignment of the call: pair_replay.append_rule puts them behind the set as the reference does
                if c["fine"]:
                    fine = ol.build_fine_index(ix, c["ivs"])
                    mf = ol.Mapper(fine)
                    p.flags |= ol.FLG_NOSHRTINFO          # initRMAPINFO, not the short form (rmap.c:2024)
                    rv, res = mf.map(seq, q, p, c["ivs"], c["prevmax"])
                    got = (pr.stage_lines(mf.dump(P["no"], name)), res, mf.cand_first, mf.stats())
                    mf.close()
                    ol.lib().or_index_free_fine(fine)
                    nfine += 1
                else:
                    rv, res = m.map(seq, q, p, c["ivs"] if c["niv"] >= 0 else None, c["prevmax"])
                    got = (pr.stage_lines(m.dump(P["no"], name)), res, m.cand_first, m.stats())
                    nrestr += c["niv"] >= 0
                try:
                    pr.check_call(c, *got)
                except AssertionError as e:
                    raise AssertionError("pair %d call %d (mate %d, %d intervals, fine %d): %s" % (P["no"], ncalls, c["mate"], c["niv"], c["fine"], str(e)[:800]))
                ncalls += 1
    finally:
        m.close()
        ol.lib().or_index_free(ix)
    assert ncalls == entry["calls"] and nfine == entry["fine_calls"] and nrestr == entry["restricted_calls"]
