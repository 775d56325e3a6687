"""The report of split reads on the CPU (`-m "not gpu"`): the alignment sets the reference's own rmapSingle leaves under
RMAPFLG_SPLIT (behind the `PE` lines of the committed `gs_*` dumps, tests/golden/make_golden_split.py) go into
smaltgpu_report_emit with SMALTGPU_OUT_SPLIT, the reads through smaltgpu_reads_parse; the text must be what `smalt map -p`
printed for those reads (`gs_*.split_cigar.out.gz`, `.split_sam.out.gz`): the primary alignment, then the best alignments of the
other read segments as partial ones (class P / SAM flag 0x100; resultSetAdd2ndaryResultsToReport, results.c:2250-2278).
Host code only; the mapping side of split reads is tests/test_oracle_split.py (CPU) and tests/test_gpu_split.py (GPU)."""
import ctypes as C
import gzip
import os

import pytest

import golden_util as gu
import split_replay as sr
from test_report import report_opts


@pytest.mark.parametrize("fmt", ["cigar", "sam"])
@pytest.mark.parametrize("entry", sr.MANIFEST, ids=[e["tag"] for e in sr.MANIFEST])
def test_split_report_matches_reference_program(entry, fmt, oracle_built, tmp_path):
    from smalt_amd import api
    L = api.lib()
    fx = sr.load_fixture(entry, tmp_path)
    text = open(fx["fq"], "rb").read()
    n = len(fx["reads"])
    by_no = {R["no"]: sr.post_state(R["post_final"]) for R in fx["dump"]}
    rows, sortr, segsrtr, segnor, dstr = [], [], [], [], bytearray()
    res_off, sort_off, seg_off = (C.c_uint64 * (n + 1))(), (C.c_uint64 * (n + 1))(), (C.c_uint64 * (n + 1))()
    qsegno, setstatus, needs = (C.c_int32 * n)(), (C.c_uint32 * n)(), (C.c_int32 * n)()
    for i in range(n):
        res_off[i], sort_off[i], seg_off[i] = len(rows), len(sortr), len(segnor)
        st = by_no.get(i)
        if not st or not st["ps"]:
            continue
        for w in st["rows"]:
            rows.append((w, len(dstr)))
            dstr += w["diffstr"]
        sortr += st["so"]
        segsrtr += st["ss"] if st["ss"] is not None else [-1] * len(st["so"])
        segnor += st["sg"] or []
        qsegno[i], setstatus[i] = st["ps"][2], st["ps"][3]
    res_off[n], sort_off[n], seg_off[n] = len(rows), len(sortr), len(segnor)
    res = (api.PostResult * max(1, len(rows)))()
    for j, (w, at) in enumerate(rows):
        r = res[j]
        r.swatscor, r.q_start, r.q_end, r.s_start, r.s_end, r.sidx = w["score"], w["q_start"], w["q_end"], w["s_start"], w["s_end"], w["sidx"]
        r.status, r.mapscor, r.prob, r.rsltx, r.qsegx, r.swrank = w["status"], w["mapscor"], w["prob"], w["rsltx"], w["qsegx"], w["swrank"]
        r.stroffs, r.strlen = at, len(w["diffstr"])
    a_sortr = (C.c_int32 * max(1, len(sortr)))(*sortr)
    a_segsrtr = (C.c_int32 * max(1, len(segsrtr)))(*segsrtr)
    a_segnor = (C.c_int32 * max(1, len(segnor)))(*segnor)
    a_dstr = (C.c_uint8 * max(1, len(dstr))).from_buffer_copy(bytes(dstr) or b"\0")
    pout = api.PostOut(n, res_off, res, a_dstr, sort_off, a_sortr, a_segsrtr, seg_off, a_segnor, qsegno, setstatus, needs)
    ro, seed = report_opts(api, ["-r", "-1", "-f", fmt] + entry["opts"].split())
    ro.outflags |= api.OUT_SPLIT
    names, seqs = fx["names"], fx["seqs"]
    sop = (C.c_uint64 * (len(seqs) + 1))()
    for i, s_ in enumerate(seqs):
        sop[i + 1] = sop[i] + len(s_)
    nm_arr = (C.c_char_p * len(names))(*[x.encode() for x in names])
    rs, rep = L.smaltgpu_reads_create(), L.smaltgpu_report_create()
    try:
        view = api.ReadsView()
        assert L.smaltgpu_reads_parse(rs, text, len(text), 1, 0, 2, C.byref(view)) == 0 and view.nreads == n
        got = b""
        txt, ln = C.c_void_p(), C.c_uint64()
        assert L.smaltgpu_report_header(rep, nm_arr, sop, len(seqs), C.byref(ro), b"smalt", b"0.7.6", 1, (C.c_char_p * 1)(b"t"), C.byref(txt), C.byref(ln)) == 0
        got += C.string_at(txt, ln.value)
        for nthreads in (1, 3):
            assert L.smaltgpu_report_emit(rep, C.byref(pout), None, C.byref(view), nm_arr, len(seqs), C.byref(ro), nthreads, C.byref(txt), C.byref(ln)) == 0, L.smaltgpu_last_error()
            body = C.string_at(txt, ln.value)
            with gzip.open(os.path.join(gu.GOLD, "%s.split_%s.out.gz" % (entry["tag"], fmt)), "rb") as g:
                exp = g.read()
            gl = [x for x in (got + body).split(b"\n") if not x.startswith(b"@PG")]
            el = [x for x in exp.split(b"\n") if not x.startswith(b"@PG")]
            assert len(gl) == len(el), (len(gl), len(el))
            for i, (x, y) in enumerate(zip(gl, el)):
                assert x == y, (i, x, y)
            assert sum(1 for x in el if x.startswith(b"cigar:P") or (b"\t" in x and not x.startswith(b"@") and int(x.split(b"\t")[1]) & 0x100)) >= 30
        # without the flag the partial alignments are not printed (the comparison is not blind to them)
        ro.outflags &= ~api.OUT_SPLIT
        assert L.smaltgpu_report_emit(rep, C.byref(pout), None, C.byref(view), nm_arr, len(seqs), C.byref(ro), 1, C.byref(txt), C.byref(ln)) == 0
        assert C.string_at(txt, ln.value) != body
    finally:
        L.smaltgpu_reads_free(rs)
        L.smaltgpu_report_free(rep)
