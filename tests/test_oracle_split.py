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
