"""BASELINE.json configs[1] at full reference size (24 x 125 Mbp, k=13 s=6) through the C ABI, checked by
size-independent properties (the oracle cannot finish this size in seconds):
  * truth recovery   -- reads are simulated from known positions: the best alignment lands there
  * score bounds     -- no score above read length x match, every reported DiffStr re-scores to its score
  * batch invariance -- the same reads mapped as one batch and as three uneven batches give identical results
  * strand symmetry  -- the reverse complement of a read finds the same best alignment on the opposite strand
                        (statistically: seeding is heuristic and sees the two orientations in a different k-mer
                        order -- repeat filter, seed budget, the strand-[0] cover deficit of segment.c:1676)
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K, S, NCHR, CHRLEN, RLEN, NREADS = 13, 6, 24, 125_000_000, 150, 24_000
NPAIRS = 12_000


def _results(out, n):
    res_off = np.ctypeslib.as_array(out.res_off, shape=(n + 1,)).copy()
    nres = int(res_off[n])
    rs = np.ctypeslib.as_array(C.cast(out.res, C.POINTER(C.c_uint8)), shape=(max(1, nres), C.sizeof(out.res._type_))).copy()
    rec = rs.view(np.dtype([("swatscor", "<i4"), ("q_start", "<u4"), ("q_end", "<u4"), ("pad", "<u4"), ("s_start", "<u8"), ("s_end", "<u8"),
                            ("sidx", "<i4"), ("reverse", "<u4"), ("stroffs", "<u4"), ("strlen", "<u4")]))[:nres, 0]
    nd = int((rec["stroffs"].astype(np.int64) + rec["strlen"]).max()) if nres else 0
    ds = np.ctypeslib.as_array(out.diffstr, shape=(max(1, nd),)).copy()
    return res_off, rec, ds


def _rescore(ds, off, length, match=1, mismatch=-2, gap_init=-4, gap_ext=-3):
    """Score of a DiffStr (diffstr.h:29-75): byte = typ << 6 | m; M(0): m + 1 matches; S(3): m matches then a substitution
    (as the last byte before the terminator: m matches only); D(1) / I(2): m matches then a one-base gap; 0 terminates."""
    score, prev_gap, i, end = 0, 0, off, off + length
    while i < end and ds[i] != 0:
        typ, m = int(ds[i]) >> 6, int(ds[i]) & 0x3F
        last = (i + 1 >= end) or ds[i + 1] == 0
        if typ == 0:
            score += (m + 1) * match
            prev_gap = 0
        elif typ == 3:
            score += m * match + (0 if last else mismatch)
            prev_gap = 0
        else:
            score += m * match + (gap_ext if (prev_gap == typ and m == 0) else gap_init)
            prev_gap = typ
        i += 1
    return score


@pytest.fixture(scope="module")
def world():
    import os
    import torch
    os.environ.setdefault("SMALTGPU_CANDS_PER_READ", "768")      # ranked-candidate pool per read (bench.py uses the same)
    from smalt_amd import api, gpuindex
    dev = torch.device("cuda", 0)
    sop = np.arange(NCHR + 1, dtype=np.int64) * CHRLEN
    ref = gpuindex.make_reference_gpu(NCHR, CHRLEN, 20261004, dev)
    packed = gpuindex.pack_reference(ref)
    idx, pos = gpuindex.build_perfect_index(ref, sop, K, S)
    reads, truth = gpuindex.make_reads_gpu(ref, sop, NREADS, RLEN, 4242)
    pr1, pr2, ptruth = gpuindex.make_pairs_gpu(ref, sop, NPAIRS, RLEN, 777)
    del ref
    desc = api.IndexDesc()
    desc.k, desc.s, desc.typ, desc.nbits_key, desc.nbits_lo = K, S, 0, 2 * K, 0
    desc.npos, desc.nwords = int(pos.numel()), 0
    desc.idx, desc.pos, desc.packed = idx.data_ptr(), pos.data_ptr(), packed.data_ptr()
    desc.nseq = NCHR
    sop_u64 = np.ascontiguousarray(sop.astype(np.uint64))
    desc.sop = sop_u64.ctypes.data
    desc.on_device = 1
    torch.cuda.synchronize()
    gix = api.Index.from_desc(desc, 0)
    mp = api.Mapper(gix, NREADS, RLEN)
    yield dict(gix=gix, mp=mp, reads=reads.cpu().numpy().reshape(NREADS, RLEN), truth=truth.cpu().numpy(), keep=(idx, pos, packed, sop_u64),
               pairs=(pr1.cpu().numpy().reshape(NPAIRS, RLEN), pr2.cpu().numpy().reshape(NPAIRS, RLEN)), ptruth=ptruth.cpu().numpy())
    mp.close()
    gix.close()


def _map(w, reads2d):
    n = reads2d.shape[0]
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(reads2d.shape[1]))
    out = w["mp"].map_batch_raw(np.ascontiguousarray(reads2d).reshape(-1), off, None, w["gix"].default_params())
    st = np.ctypeslib.as_array(C.cast(out.stat, C.POINTER(C.c_uint32)), shape=(n, 10)).copy()
    return _results(out, n) + (st,)


def test_truth_recovery_and_score_bounds(world):
    res_off, rec, ds, st = _map(world, world["reads"])
    assert (st[:, 6] == 0).all()                                   # no per-read error
    assert (rec["swatscor"] <= RLEN).all() and (rec["swatscor"] >= 1).all()
    hit = 0
    for i in range(NREADS):
        r = rec[res_off[i]:res_off[i + 1]]
        assert len(r) > 0
        b = r[np.argmax(r["swatscor"])]
        seq, pos, strand = world["truth"][i]
        if b["sidx"] == seq and abs(int(b["s_start"]) - 1 - int(pos)) <= 12 and (int(b["reverse"]) & 1) == int(strand):
            hit += 1
    assert hit >= 0.97 * NREADS, hit                              # repeats (15 % of the reference) may place a read elsewhere
    rng = np.random.default_rng(1)
    for j in rng.integers(0, len(rec), size=3000):                 # DiffStr re-scores to the reported score
        assert _rescore(ds, int(rec["stroffs"][j]), int(rec["strlen"][j])) == int(rec["swatscor"][j]), j


def test_batch_split_invariance(world):
    whole = _map(world, world["reads"])
    cuts = [0, 5000, 5001, 17011, NREADS]
    k = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        res_off, rec, ds, st = _map(world, world["reads"][a:b])
        for i in range(b - a):
            x = rec[res_off[i]:res_off[i + 1]]
            y = whole[1][whole[0][a + i]:whole[0][a + i + 1]]
            assert len(x) == len(y), (a, i)
            for f in ("swatscor", "q_start", "q_end", "s_start", "s_end", "sidx", "reverse", "strlen"):
                assert (x[f] == y[f]).all(), (a, i, f)
            k += len(x)
        assert (st[:, :6] == whole[3][a:b, :6]).all()
    assert k == len(whole[1])


def test_strand_symmetry(world):
    sub = world["reads"][:6000]
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    rc = comp[sub][:, ::-1]
    f = _map(world, sub)
    r = _map(world, rc)
    same = same_score = 0
    for i in range(sub.shape[0]):
        x = f[1][f[0][i]:f[0][i + 1]]
        y = r[1][r[0][i]:r[0][i + 1]]
        bx, by = x[np.argmax(x["swatscor"])], y[np.argmax(y["swatscor"])]
        same_score += int(bx["swatscor"] == by["swatscor"])
        if bx["sidx"] == by["sidx"] and bx["s_start"] == by["s_start"] and bx["s_end"] == by["s_end"] and (bx["reverse"] & 1) != (by["reverse"] & 1):
            same += 1
    assert same_score >= 0.97 * sub.shape[0], same_score
    assert same >= 0.95 * sub.shape[0], same                      # ties between repeat copies may resolve differently

def _pair_lines(w, r1, r2, nthreads=8):
    """One block of pairs through smaltgpu_map_pairs and smaltgpu_report_emit_pairs (CIGAR lines, -i 500, no random draws):
    -> (lines, pair info array)"""
    from smalt_amd import api
    L = api.lib()
    n = r1.shape[0]
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(r1.shape[1])
    po = api.PairOpts(0, 500, api.LIB_PE, 0, nthreads)
    b1, b2 = np.ascontiguousarray(r1).reshape(-1), np.ascontiguousarray(r2).reshape(-1)
    h, info, calls, ms = w["mp"].map_pairs_raw(b1, off, None, b2, off, None, w["gix"].default_params(), po)
    views, keep = [], []
    for which, rd in ((1, r1), (2, r2)):
        text = b"".join(b"@p%d/%d\n" % (i, which) + rd[i].tobytes() + b"\n+\n" + b"I" * rd.shape[1] + b"\n" for i in range(n))
        rs = L.smaltgpu_reads_create()
        v = api.ReadsView()
        assert L.smaltgpu_reads_parse(rs, text, len(text), 1, 0, nthreads, C.byref(v)) == 0, L.smaltgpu_last_error()
        views.append(v)
        keep.append((rs, text))
    ro = api.ReportOpts()
    ro.format = api.FMT_CIGAR
    ro.min_swscor = 18
    ro.outflags = api.OUT_BEST | api.OUT_SINGLE                  # `-r -1`: ambiguous pairs are reported as such, no draws
    names = (C.c_char_p * NCHR)(*[b"chr%d" % (i + 1) for i in range(NCHR)])
    rep = L.smaltgpu_report_create()
    txt, ln = C.c_void_p(), C.c_uint64()
    assert L.smaltgpu_report_emit_pairs(rep, h, C.byref(views[0]), C.byref(views[1]), names, NCHR, C.byref(ro), C.byref(po), nthreads, C.byref(txt), C.byref(ln)) == 0, L.smaltgpu_last_error()
    lines = C.string_at(txt, ln.value).decode().split("\n")
    info = info.copy()
    L.smaltgpu_report_free(rep)
    L.smaltgpu_pairs_free(h)
    for rs, _ in keep:
        L.smaltgpu_reads_free(rs)
    return [x for x in lines if x], info


def test_pairs_at_full_reference_size(world):
    """BASELINE configs[2] shape at full reference size through smaltgpu_map_pairs + smaltgpu_report_emit_pairs, by properties:
      * truth recovery  -- pairs are simulated as fragments N(300,30) at known positions: both mates land there, labelled as a
                           proper pair (class A of the CIGAR format), with the fragment's length between them
      * block invariance -- the same pairs as one block and as three uneven blocks print the same lines
      * mate symmetry   -- with the two files swapped every pair keeps its two placements (the mates change roles)"""
    r1, r2 = world["pairs"]
    lines, info = _pair_lines(world, r1, r2)
    assert len(lines) == 2 * NPAIRS
    assert (np.ascontiguousarray(info[:, 2:6]).view(np.uint16).reshape(NPAIRS, 2) > 0).all()          # both mates have alignments
    good = 0
    for p in range(NPAIRS):
        f1, f2 = lines[2 * p].split(), lines[2 * p + 1].split()
        assert f1[0].startswith("cigar:") and f1[1] == "p%d/1" % p and f2[1] == "p%d/2" % p
        seq, start, strand, flen = (int(x) for x in world["ptruth"][p])
        cls = f1[0].split(":")[1]
        lo = min(int(f1[6]), int(f1[7]), int(f2[6]), int(f2[7]))
        hi = max(int(f1[6]), int(f1[7]), int(f2[6]), int(f2[7]))
        if cls == "A" and f1[5] == "chr%d" % (seq + 1) == f2[5] and abs(lo - 1 - start) <= 12 and abs((hi - lo + 1) - flen) <= 24:
            good += 1
    assert good >= 0.95 * NPAIRS, good                         # repeats (15 % of the reference) place some pairs elsewhere
    cuts = [0, 3000, 3001, 9500, NPAIRS]
    k = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        part, _ = _pair_lines(world, r1[a:b], r2[a:b])
        for i, ln in enumerate(part):
            want = lines[2 * a + i].split()
            got = ln.split()
            got[1] = want[1]                                    # the names carry the pair's number within the block
            assert got == want, (a, i)
            k += 1
    assert k == 2 * NPAIRS
    swapped, _ = _pair_lines(world, r2[:4000], r1[:4000])
    same = 0
    for p in range(4000):
        a1, a2 = lines[2 * p].split(), lines[2 * p + 1].split()
        b1, b2 = swapped[2 * p].split(), swapped[2 * p + 1].split()
        if a1[5:8] == b2[5:8] and a2[5:8] == b1[5:8]:
            same += 1
    assert same >= 0.97 * 4000, same                          # which mate is mapped first can change what a repeat pair finds


def test_bench_two_ranks_on_one_device():
    """`python bench.py --gpus 2` on the one-GPU box: the program starts its two ranks itself, they share the device (so the
    collectives run over gloo; with a GPU per rank the backend is nccl = RCCL), rank 0 builds the index and broadcasts the image,
    the reads of the job are dealt in guided pieces from one cursor, and the reductions give one JSON line.  Small genome so
    that the two processes' images fit beside each other."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--reads", "131072", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--no-host-buffers", "--nchr", "4", "--chr-mbp", "25"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["job_reads_per_step"] == 262144
    lo, hi = line["config"]["reads_taken_min_max_per_rank"]
    assert lo > 0 and abs(lo + hi - 262144) < 1            # the two ranks' pieces add up to the job
    assert line["config"]["mapped_fraction"] > 0.999
    assert line["config"]["index_broadcast_ms"] > 0 and set(line["config"]["index_broadcast_by_array"]) >= {"idx", "pos", "packed"}
