"""CPU oracle vs the committed fixtures generated from the unmodified reference
(tests/golden/make_golden.py): index files byte-identical to `smalt index`, and every stage of
the per-read path (seeds, hit lists, candidates, bands, scores, alignments) line-identical to
the reference's own functions."""
import os
import subprocess

import pytest

import golden_util as gu


@pytest.mark.parametrize("entry", gu.MANIFEST, ids=[e["tag"] for e in gu.MANIFEST])
def test_oracle_matches_reference_dump(entry, oracle_built, tmp_path):
    fx = gu.unpack(entry, tmp_path)
    assert gu.md5(fx["prefix"] + ".sma") == entry["sma_md5"]
    assert gu.md5(fx["prefix"] + ".smi") == entry["smi_md5"]
    out = subprocess.run([os.path.join(oracle_built, "ordump")] + entry["opts"].split() + [fx["prefix"], fx["fq"]],
                         check=True, capture_output=True, text=True).stdout
    if out != fx["expected"]:
        a, b = out.split("\n"), fx["expected"].split("\n")
        for i, (x, y) in enumerate(zip(a, b)):
            assert x == y, "line %d" % (i + 1)
        assert len(a) == len(b)
