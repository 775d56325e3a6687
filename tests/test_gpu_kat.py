"""The reference-held known-answer vectors of test/bam_cigar_test.py (tests/golden/kat_bam_cigar.json: k=7 s=1, exact
CIGAR, extended CIGAR and NM) through libsmaltgpu: index built by the library's own builder, reads mapped by
api.Mapper.  tests/test_oracle_kat.py runs the same vectors through the CPU oracle."""
import json
import os

import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("how", ["built", "loaded"])
def test_bam_cigar_kat_on_the_gpu(how, oracle_built, tmp_path):
    from smalt_amd import api
    doc = json.load(open(os.path.join(HERE, "golden", "kat_bam_cigar.json")))
    seqs = [s.upper().encode() for s in doc["refseq"]]
    names = ["REF_%d" % i for i in range(len(seqs))]
    gix = api.Index.build(seqs, names, doc["k"], doc["s"], 0)
    if how == "loaded":                 # the same through the index files (written by the library, read back)
        pre = os.path.join(str(tmp_path), "kat")
        gix.save(pre)
        gix.close()
        gix = api.Index.load(pre, 0)
    par = gix.default_params()
    rds = [r["seq"].upper().encode() for r in doc["reads"] + doc["pair_reads"]]
    mp = api.Mapper(gix, 16, 64)
    try:
        res, stats = mp.map_batch(rds, None, par)
        for rd, rr, seq in zip(doc["reads"] + doc["pair_reads"], res, rds):
            assert rr
            top = max(r["score"] for r in rr)
            best = [r for r in rr if r["score"] == top]
            assert len(best) == 1
            b = best[0]
            cig, nm = ol.diffstr_to_cigar(b["diffstr"], b["q_start"], b["q_end"], len(seq), b["reverse"])
            cigx, _ = ol.diffstr_to_cigar(b["diffstr"], b["q_start"], b["q_end"], len(seq), b["reverse"], ext=True)
            assert (cig, cigx, nm) == (rd["cigar"], rd["cigar_x"], rd["nm"]), (rd["seq"], cig, cigx, nm)
    finally:
        mp.close()
        gix.close()
