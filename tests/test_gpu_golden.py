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


@pytest.mark.parametrize("history,batches", [(True, 1), (True, 3), (False, 1)], ids=["serial-order", "serial-order-3-calls", "stateless"])
def test_serial_order_mode_follows_the_hit_list_of_a_serial_run(history, batches, oracle_built, tmp_path):
    """tests/golden/make_golden_history.py: reads of 100 bases that reach the allocation boundary of their hit list (hashhit.c:1497)
    unless a longer read was mapped before them (initHitList only ever grows the list, hashhit.c:1280).  With
    smaltgpu_mapper_set_history the mapper reproduces the reference run serially over the file -- also when the file is mapped in
    several calls, the longest length is carried between them; without it every read is mapped as the first read of a run
    (second dump of the fixture: the reference run on each read alone)."""
    import gzip
    import json
    import os

    from smalt_amd import api
    entry = json.load(open(os.path.join(gu.GOLD, "manifest_history.json")))
    fx = gu.unpack(entry, tmp_path)
    expected = fx["expected"] if history else gzip.open(os.path.join(gu.GOLD, entry["tag"] + ".refdump_fresh.txt.gz"), "rt").read()
    reads = gu.read_fastq(fx["fq"])
    ix = api.Index.load(fx["prefix"], 0)
    mp = api.Mapper(ix, 2048, max(len(r[1]) for r in reads))      # pools for 2048 average reads: the repeat reads have thousands of candidates each
    got = []
    try:
        mp.set_debug(2)
        if history:
            mp.set_history(True)
        par = params_from_opts(ix, entry["opts"])
        step = (len(reads) + batches - 1) // batches
        for b0 in range(0, len(reads), step):
            part = reads[b0:b0 + step]
            res, stats = mp.map_batch([r[1] for r in part], [r[2] for r in part], par)
            assert all(s["err"] == 0 for s in stats)
            got += [mp.dump_read(i, part[i][0]).replace("READ %d " % i, "READ %d " % (b0 + i), 1) for i in range(len(part))]
    finally:
        mp.close()
        ix.close()
    a, b = "".join(got).split("\n"), expected.split("\n")
    for i, (x, y) in enumerate(zip(a, b)):
        assert x == y, "line %d" % (i + 1)
    assert len(a) == len(b)
