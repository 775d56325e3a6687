"""K2a kernel (register-tiled, DPP) against the oracle's textbook Gotoh maximum on random
(query, window) pairs in every tile geometry, with indels, Ns, and scores past 8-bit range."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _pairs(rng, n, qlen_lo, qlen_hi, with_n):
    qs, ws = [], []
    for _ in range(n):
        ql = int(rng.integers(qlen_lo, qlen_hi + 1))
        q = rng.integers(0, 4, size=ql, dtype=np.uint8)
        # window: mutated copy of the query embedded in random flanks, or unrelated
        if rng.random() < 0.8:
            core = q.copy()
            m = rng.random(ql) < rng.choice([0.0, 0.02, 0.1, 0.3])
            core = np.where(m, (core + rng.integers(1, 4, size=ql)) & 3, core).astype(np.uint8)
            if rng.random() < 0.5 and ql > 20:
                p = int(rng.integers(5, ql - 5))
                core = np.concatenate([core[:p], core[p + int(rng.integers(1, 4)):]]) if rng.random() < 0.5 else \
                    np.concatenate([core[:p], rng.integers(0, 4, size=int(rng.integers(1, 4)), dtype=np.uint8), core[p:]])
            w = np.concatenate([rng.integers(0, 4, size=int(rng.integers(0, 40)), dtype=np.uint8), core,
                                rng.integers(0, 4, size=int(rng.integers(0, 40)), dtype=np.uint8)])
        else:
            w = rng.integers(0, 4, size=int(rng.integers(1, ql + 80)), dtype=np.uint8)
        if with_n:
            if rng.random() < 0.3:
                q[int(rng.integers(0, ql))] = 5
            if rng.random() < 0.3:
                w[int(rng.integers(0, len(w)))] = 5
        qs.append(q.tobytes())
        ws.append(w.tobytes())
    return qs, ws


@pytest.mark.parametrize("qlo,qhi", [(32, 64), (65, 104), (105, 152), (153, 160), (161, 256), (257, 512)])
def test_sw_full_kernel_matches_oracle(qlo, qhi, oracle_built):
    from smalt_amd import api
    rng = np.random.default_rng(qlo * 7919 + qhi)
    seqs = [bytes(rng.choice(list(b"ACGT"), size=4000).astype(np.uint8))]
    oix = ol.build_index(seqs, ["s"], 11, 3)
    import os, tempfile
    with tempfile.TemporaryDirectory() as tmp:
        pre = os.path.join(tmp, "x")
        ol.lib().or_index_write(oix, pre.encode())
        gix = api.Index.load(pre, 0)
    mp = api.Mapper(gix, 16, 512)
    par = gix.default_params()
    M = (C.c_int8 * 64)()
    ol.lib().or_score_matrix(M, 1, -2)
    try:
        for with_n in (False, True):
            qs, ws = _pairs(rng, 700, qlo, qhi, with_n)
            got = mp.sw_full_batch(qs, ws, par)
            got16 = mp.sw_full_batch(qs, ws, par, packed16=True)
            for i, (q, w) in enumerate(zip(qs, ws)):
                exp = ol.lib().or_sw_full(q, len(q), w, len(w), M, -4, -3)
                assert got[i] == exp, (i, len(q), len(w), got[i], exp)
                if any(c >= 4 for c in q):
                    assert got16[i] == -2, (i, got16[i])       # non-ACGT query: left to the 32-bit kernel
                else:
                    assert got16[i] == exp, (i, len(q), len(w), got16[i], exp)
    finally:
        mp.close()
        gix.close()
        ol.lib().or_index_free(oix)


@pytest.mark.parametrize("pen", [(2, -3, -5, -2), (5, -4, -6, -1), (1, -1, -2, -1), (3, -6, -8, -4),       # half-float form
                                 (9, -7, -9, -3), (3, -9, -4, -3), (15, -14, -20, -10), (13, -2, -4, -3)],  # integer form (low byte / range)
                         ids=lambda v: "m%d_x%d_g%d_e%d" % (v[0], -v[1], -v[2], -v[3]))
def test_sw_packed_kernel_penalty_sets(pen, oracle_built):
    """The packed K2a kernel picks its form from the penalties: half floats when match and mismatch are half floats with a
    zero low byte and match x tile columns <= 2040, the integer form otherwise.  Both against the oracle."""
    from smalt_amd import api
    match, mismatch, gi, ge = pen
    rng = np.random.default_rng(match * 1000 - mismatch * 10 - gi)
    seqs = [bytes(rng.choice(list(b"ACGT"), size=4000).astype(np.uint8))]
    oix = ol.build_index(seqs, ["s"], 11, 3)
    import os, tempfile
    with tempfile.TemporaryDirectory() as tmp:
        pre = os.path.join(tmp, "x")
        ol.lib().or_index_write(oix, pre.encode())
        gix = api.Index.load(pre, 0)
    mp = api.Mapper(gix, 16, 512)
    par = gix.default_params()
    par.match, par.mismatch, par.gap_init, par.gap_ext = match, mismatch, gi, ge
    M = (C.c_int8 * 64)()
    ol.lib().or_score_matrix(M, match, mismatch)
    try:
        for qlo, qhi in ((40, 64), (105, 152), (161, 256)):
            qs, ws = _pairs(rng, 400, qlo, qhi, False)
            got16 = mp.sw_full_batch(qs, ws, par, packed16=True)
            got32 = mp.sw_full_batch(qs, ws, par)
            for i, (q, w) in enumerate(zip(qs, ws)):
                exp = ol.lib().or_sw_full(q, len(q), w, len(w), M, gi, ge)
                assert got16[i] == exp, (i, len(q), len(w), got16[i], exp)
                assert got32[i] == exp, (i, len(q), len(w), got32[i], exp)
    finally:
        mp.close()
        gix.close()
        ol.lib().or_index_free(oix)


@pytest.mark.parametrize("qlo,qhi,ntask", [(513, 1024, 60), (1025, 2300, 40), (40, 300, 60)])
def test_sw_strip_kernel_matches_oracle(qlo, qhi, ntask, oracle_built):
    """K2a beyond the register tiling: strips of 1024 read columns with the boundary column handed from strip to strip
    (1, 2 and 3 strips); the third case forces short reads with windows > 1016 through the same kernel."""
    from smalt_amd import api
    rng = np.random.default_rng(qlo * 31 + qhi)
    seqs = [bytes(rng.choice(list(b"ACGT"), size=4000).astype(np.uint8))]
    oix = ol.build_index(seqs, ["s"], 11, 3)
    import os, tempfile
    with tempfile.TemporaryDirectory() as tmp:
        pre = os.path.join(tmp, "x")
        ol.lib().or_index_write(oix, pre.encode())
        gix = api.Index.load(pre, 0)
    mp = api.Mapper(gix, 16, 2400)
    par = gix.default_params()
    M = (C.c_int8 * 64)()
    ol.lib().or_score_matrix(M, 1, -2)
    try:
        qs, ws = _pairs(rng, ntask, qlo, qhi, True)
        if qhi <= 512:          # long windows for short reads
            ws = [w + bytes(rng.integers(0, 4, size=1100, dtype=np.uint8)) for w in ws]
        got = mp.sw_full_batch(qs, ws, par)
        for i, (q, w) in enumerate(zip(qs, ws)):
            exp = ol.lib().or_sw_full(q, len(q), w, len(w), M, -4, -3)
            assert got[i] == exp, (i, len(q), len(w), got[i], exp)
    finally:
        mp.close()
        gix.close()
        ol.lib().or_index_free(oix)


def test_packed_max3_orders_u16_bit_patterns_like_integers(tmp_path):
    """The packed kernels use gfx950's v_pk_maximum3_f16 as an integer maximum of three on u16 bit patterns below 0x7C00
    (tools/pk_max3_check.hip: 10^6 random triples incl. the half-float denormal range, bit for bit)."""
    import os, shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "pk_max3_check")
    subprocess.run([hipcc, "-w", "-O2", "--offload-arch=gfx950", "-o", exe, os.path.join(root, "tools", "pk_max3_check.hip")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "bad 0 of" in out
