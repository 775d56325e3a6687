"""GPU parity proper: every stage of the HIP path, through the C ABI, against the dumps of the
unmodified reference committed under tests/golden/ (bit-exact: line-identical text)."""
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


def params_from_opts(ix, opts):
    from smalt_amd import api
    p = ix.default_params()
    o = opts.split()
    i = 0
    while i < len(o):
        if o[i] == "-d":
            p.min_swatscor_below_max = int(o[i + 1]); i += 2
            if p.min_swatscor_below_max != 0:
                p.rmapflg &= ~api.FLG_BEST
        elif o[i] == "-q":
            p.min_basqval = int(o[i + 1]); i += 2
        elif o[i] == "-m":
            p.min_swatscor = int(o[i + 1]); i += 2
        elif o[i] == "-c":
            v = float(o[i + 1]); i += 2
            if v < 1.01:
                p.min_cover_frac = v          # per read: (uint32_t)(v * read length), smalt.c:1113-1122
            else:
                p.min_cover = int(v)
        elif o[i] == "-x":
            p.rmapflg |= api.FLG_NOSHRTINFO | api.FLG_SENSITIVE; i += 1
        else:
            raise ValueError(o[i])
    return p


@pytest.mark.parametrize("entry", gu.MANIFEST, ids=[e["tag"] for e in gu.MANIFEST])
def test_gpu_matches_reference_dump(entry, oracle_built, tmp_path):
    from smalt_amd import api
    fx = gu.unpack(entry, tmp_path)
    reads = gu.read_fastq(fx["fq"])
    ix = api.Index.load(fx["prefix"], 0)
    mp = api.Mapper(ix, len(reads), max(len(r[1]) for r in reads))
    try:
        mp.set_debug(2)
        res, stats = mp.map_batch([r[1] for r in reads], [r[2] for r in reads], params_from_opts(ix, entry["opts"]))
        assert all(s["err"] == 0 for s in stats)
        got = "".join(mp.dump_read(i, reads[i][0]) for i in range(len(reads)))
    finally:
        mp.close()
        ix.close()
    a, b = got.split("\n"), fx["expected"].split("\n")
    for i, (x, y) in enumerate(zip(a, b)):
        assert x == y, "line %d" % (i + 1)
    assert len(a) == len(b)
