"""libsmaltgpu vs the committed PAIRED fixtures (tests/golden/make_golden_pairs.py: dumps of the reference's own rmapPair,
rmap.c:1744, one block per mapSingleRead call it makes).  Every call of every pair is replayed through the C ABI with the
call's recorded arguments -- batched the way a binding would batch the rounds of rmapPair: all unrestricted calls of the
fixture in one smaltgpu_map_batch_ctx call, all interval-restricted calls in a second, all calls over the on-the-fly k=5
index in a third -- and must reproduce the reference's stage state line by line, its alignments (appended to a non-empty
ResultSet as results.c:1906-1935 does) and the result set's running score maxima.  tests/test_oracle_pairs.py runs the same
fixtures through the CPU oracle."""
import json
import os

import pytest

import golden_util as gu
import pair_replay as pr

pytestmark = pytest.mark.gpu
MANIFEST = json.load(open(os.path.join(gu.GOLD, "manifest_pairs.json")))


@pytest.mark.parametrize("entry", MANIFEST, ids=[e["tag"] for e in MANIFEST])
def test_gpu_replays_every_call_of_rmappair(entry, oracle_built, tmp_path):
    from smalt_amd import api
    fx = pr.load_fixture(entry, tmp_path)
    gix = api.Index.load(fx["prefix"], 0)
    calls = [(P, c) for P in fx["pairs"] for c in P["calls"]]
    rounds = {"plain": [pc for pc in calls if pc[1]["niv"] < 0], "restricted": [pc for pc in calls if pc[1]["niv"] >= 0 and not pc[1]["fine"]],
              "fine": [pc for pc in calls if pc[1]["fine"]]}
    assert len(rounds["fine"]) == entry["fine_calls"] and len(rounds["restricted"]) == entry["restricted_calls"]
    maxlen = max(len(r[1]) for r in fx["reads1"] + fx["reads2"])
    ndone = 0
    try:
        for what, lst in rounds.items():
            if not lst:
                continue
            keys = sorted({(c["mincov"], c["flags"], c["belowmax"]) for _, c in lst})      # batch-wide parameters
            for key in keys:
                sub = [(P, c) for P, c in lst if (c["mincov"], c["flags"], c["belowmax"]) == key]
                rd = [(fx["reads2"] if c["mate"] else fx["reads1"])[P["no"]] for P, c in sub]
                par = gix.default_params()
                par.min_cover, par.min_swatscor_below_max, par.min_basqval = key[0], key[2], fx["min_basq"]
                par.rmapflg = key[1] & (api.FLG_BEST | api.FLG_SEQBYSEQ | api.FLG_NOSHRTINFO | api.FLG_SENSITIVE)
                mp = api.Mapper(gix, len(sub), maxlen)
                mp.set_debug(1)
                try:
                    res, stats, cf = mp.map_batch_ctx([r[1] for r in rd], [r[2] for r in rd], par,
                                                      intervals=None if what == "plain" else [c["ivs"] for _, c in sub],
                                                      min_swatscor=[c["minscor"] for _, c in sub], prev_max=[c["prevmax"] for _, c in sub],
                                                      fine_index=(what == "fine"), raw_alignments=True)     # every alignment: pair_replay.append_rule puts them behind the set
                    for i, (P, c) in enumerate(sub):
                        st = stats[i]
                        assert st["err"] == 0
                        lines = pr.stage_lines(mp.dump_read(i, rd[i][0]))
                        lines[0] = lines[0].replace("READ %d " % i, "READ %d " % P["no"], 1)
                        try:
                            pr.check_call(c, lines, res[i], cf[i], (st["swmax"], st["sw2nd"], st["nseg"], st["nseg_tot"], st["nhit"], st["nhit_tot"]))
                        except AssertionError as e:
                            raise AssertionError("%s round, pair %d (mate %d, %d intervals): %s" % (what, P["no"], c["mate"], c["niv"], str(e)[:800]))
                        # mapSingleRead leaves before the traceback pass iff the best first-pass score is below 1 (rmap.c:1376)
                        assert (st["max1"] >= 1) or not res[i]
                        ndone += 1
                finally:
                    mp.close()
        assert ndone == entry["calls"]
        # the rare-mate decision (rmap.c:1866-1870): hit totals of both mates of every pair vs the order the reference took
        mp = api.Mapper(gix, 2 * len(fx["pairs"]), maxlen)
        try:
            par = gix.default_params()
            par.min_basqval = fx["min_basq"]
            rds = [x for P in fx["pairs"] for x in (fx["reads1"][P["no"]], fx["reads2"][P["no"]])]
            tot = mp.hit_totals([r[1] for r in rds], [r[2] for r in rds], par)
            k = gix.info().k
            for j, P in enumerate(fx["pairs"]):
                l1, l2 = len(rds[2 * j][1]), len(rds[2 * j + 1][1])
                if l1 < k or l2 < k or not P["calls"]:
                    continue                                  # a mate shorter than the word length: mapped alone first (rmap.c:1836-1864)
                first = P["calls"][0]["mate"]
                assert first == (1 if tot[2 * j] > tot[2 * j + 1] else 0), (P["no"], tot[2 * j], tot[2 * j + 1], first)
        finally:
            mp.close()
    finally:
        gix.close()
