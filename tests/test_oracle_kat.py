"""Pin the CPU oracle with the known-answer vectors the reference's own test suite holds for
this path (test/bam_cigar_test.py: exact CIGAR, extended CIGAR and NM for k=7, s=1)."""
import json
import os

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))


def _best(results):
    top = max(r["score"] for r in results)
    best = [r for r in results if r["score"] == top]
    assert len(best) == 1
    return best[0]


def test_bam_cigar_kat(oracle_built):
    doc = json.load(open(os.path.join(HERE, "golden", "kat_bam_cigar.json")))
    seqs = [s.encode() for s in doc["refseq"]]
    ix = ol.build_index(seqs, ["REF_%d" % i for i in range(len(seqs))], doc["k"], doc["s"])
    par = ol.default_params(ix)
    m = ol.Mapper(ix)
    try:
        for rd in doc["reads"] + doc["pair_reads"]:
            seq = rd["seq"].encode()
            rv, res = m.map(seq, None, par)
            assert rv == 0 and res
            b = _best(res)
            cig, nm = ol.diffstr_to_cigar(b["diffstr"], b["q_start"], b["q_end"], len(seq), b["reverse"])
            cigx, _ = ol.diffstr_to_cigar(b["diffstr"], b["q_start"], b["q_end"], len(seq), b["reverse"], ext=True)
            assert cig == rd["cigar"], (rd["seq"], cig)
            assert cigx == rd["cigar_x"], (rd["seq"], cigx)
            assert nm == rd["nm"]
    finally:
        m.close()
        ol.lib().or_index_free(ix)
