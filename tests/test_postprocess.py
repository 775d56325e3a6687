"""Result post-processing of libsmaltgpu against the reference: the committed `*.post.txt.gz` fixtures are
`oracle/_ref/refdump -n -p` output for the golden inputs (tests/golden/make_golden.py) -- per read the raw alignments
(`RW`, `RX`, `RC`: what the GPU path delivers) and the state the reference's resultSetSortAndAssignSequence
(results.c:2022) leaves behind: per-alignment status, mapping quality, probability, sequence assignment, segment and rank
(`RF`), the sorted array (`SO`), the per-segment array and its bounds (`SS`, `SG`), set status and segment count (`PS`).
smaltgpu_postprocess is host code, so this test needs no GPU: it feeds the raw alignments in and compares everything,
mapping probabilities to the last bit.  The fixture with 600 reference sequences (concatenated mode) holds alignments that
run across sequence junctions: their fragments (splitMultiSpan, results.c:1472) must come out with the reference's
coordinates, scores and alignment strings."""
import ctypes as C
import gzip
import os

import pytest

import golden_util as gu

MASK = ~(0x10 | 0x20 | 0x200)          # output filters and the report set these bits later (rd_results.c)


def _blocks(tag):
    cur, pend = None, []
    with gzip.open(os.path.join(gu.GOLD, tag + ".post.txt.gz"), "rt") as g:
        for ln in g:
            f = ln.split()
            if not f:
                continue
            if f[0] == "READ":
                if cur:
                    yield cur
                cur = dict(name=f[2], rs=pend, rc=[], rf=[], so=None, ss=None, sg=None, ps=None, rx=None)
                pend = []
            elif f[0] == "RW":                 # the alignments before the post-processing (printed ahead of their READ line)
                pend.append(f)
            elif f[0] == "RC":
                cur["rc"].append(int(f[-1]))
            elif f[0] == "RX":
                cur["rx"] = [int(x) for x in f[1:]]
            elif f[0] == "PS":
                cur["ps"] = [int(x) for x in f[1:]]
            elif f[0] == "RF":
                cur["rf"].append(f)
            elif f[0] in ("SO", "SS", "SG"):
                cur[f[0].lower()] = [int(x) for x in f[1:]]
    if cur:
        yield cur


@pytest.mark.parametrize("entry", gu.MANIFEST_ALL, ids=[e["tag"] for e in gu.MANIFEST_ALL])
def test_postprocess_matches_reference(entry, oracle_built, tmp_path):
    from smalt_amd import api
    L = api.lib()
    fx = gu.unpack(entry, tmp_path)
    reads = {r[0]: r for r in gu.read_fastq(fx["fq"])}
    blocks = list(_blocks(entry["tag"]))
    assert len(blocks) == len(reads)
    n = len(blocks)
    nres = sum(len(b["rs"]) for b in blocks)
    res = (api.Result * max(1, nres))()
    stat = (api.ReadStat * n)()
    res_off = (C.c_uint64 * (n + 1))()
    read_off = (C.c_uint64 * (n + 1))()
    dstr, quals, bases = bytearray(), bytearray(), bytearray()
    j = 0
    for i, b in enumerate(blocks):
        res_off[i] = j
        read_off[i] = len(quals)
        quals += reads[b["name"]][2]
        bases += reads[b["name"]][1]
        for f in b["rs"]:
            r = res[j]
            r.reverse = 1 if f[2] == "R" else 0
            r.swatscor, r.q_start, r.q_end, r.s_start, r.s_end, r.sidx = int(f[3]), int(f[4]), int(f[5]), int(f[6]), int(f[7]), int(f[8])
            d = bytes.fromhex(f[9])
            r.stroffs, r.strlen = len(dstr), len(d)
            dstr += d
            j += 1
        st = stat[i]
        st.swatscor_max, st.swatscor_2ndmax, st.n_ali_done, st.n_ali_tot, st.n_hits_used, st.n_hits_tot = b["rx"][1:7]
        st.nres = len(b["rs"])
        st.max1scor = max(b["rc"]) if b["rc"] else 0          # best first-pass score: mapSingleRead sorts only if it is >= 1 (rmap.c:1376)
    res_off[n] = j
    read_off[n] = len(quals)
    dbuf = (C.c_uint8 * max(1, len(dstr))).from_buffer_copy(bytes(dstr) or b"\0")
    raw = api.BatchOut(n, res_off, res, dbuf, stat)
    names, seqs = fx["names"], fx["seqs"]
    sop = (C.c_uint64 * (len(seqs) + 1))()
    for i, s_ in enumerate(seqs):
        sop[i + 1] = sop[i] + len(s_)
    post = L.smaltgpu_post_create()
    out = api.PostOut()
    qb = (C.c_uint8 * max(1, len(quals))).from_buffer_copy(bytes(quals) or b"\0")
    bb = (C.c_uint8 * max(1, len(bases))).from_buffer_copy(bytes(bases) or b"\0")
    import oracle_lib as ol
    oix = ol.lib().or_index_read(fx["prefix"].encode())          # host copy of the packed reference (the .sma file's words)
    par = api.Params()
    par.match, par.mismatch, par.gap_init, par.gap_ext = 1, -2, -4, -3
    try:
        for nthreads, split in ((1, True), (3, True), (2, False)):
            rv = L.smaltgpu_postprocess(post, sop, len(seqs), C.byref(raw), bb if split else None, qb, read_off,
                                        C.cast(oix.contents.packed, C.c_void_p) if split else None, C.byref(par) if split else None, nthreads, C.byref(out))
            assert rv == 0
            nspan = 0
            for i, b in enumerate(blocks):
                a, e = out.res_off[i], out.res_off[i + 1]
                if out.needs_reference[i]:                    # without the reference: an alignment across a sequence junction is left to the caller
                    assert not split
                    nspan += 1
                    assert len(b["rf"]) > len(b["rs"])
                    continue
                assert b["ps"][0] == e - a and b["ps"][2] == out.qsegno[i] and b["ps"][3] == out.setstatus[i], (i, b["ps"], out.qsegno[i], out.setstatus[i])
                for k_, f in enumerate(b["rf"]):
                    r = out.res[a + k_]
                    got = (r.status & MASK, r.swatscor, r.mapscor, repr(r.prob), r.q_start, r.q_end, r.s_start, r.s_end, r.sidx, r.rsltx, r.qsegx, r.swrank,
                           bytes(out.diffstr[r.stroffs:r.stroffs + r.strlen]).hex())
                    exp = (int(f[2]) & MASK, int(f[3]), int(f[4]), repr(float(f[5])), int(f[6]), int(f[7]), int(f[8]), int(f[9]), int(f[10]), int(f[11]), int(f[12]), int(f[13]),
                           f[14] if len(f) > 14 else "")
                    assert got == exp, (i, k_, got, exp)
                so = [out.sortr[x] for x in range(out.sort_off[i], out.sort_off[i + 1])]
                assert so == (b["so"] or []), (i, so, b["so"])
                if b["ss"] is not None:
                    assert [out.segsrtr[x] for x in range(out.sort_off[i], out.sort_off[i + 1])] == b["ss"], i
                    assert [out.segnor[x] for x in range(out.seg_off[i], out.seg_off[i + 1])] == b["sg"], i
            if not split:
                assert nspan == sum(1 for b in blocks if len(b["rf"]) > len(b["rs"]))
                assert nspan >= (30 if entry["tag"] == "g_k11s4_cat" else 0)      # that fixture holds alignments across junctions
    finally:
        L.smaltgpu_post_free(post)
        ol.lib().or_index_free(oix)
