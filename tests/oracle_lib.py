"""ctypes bindings of oracle/liboracle.so -- the CPU checker.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
FLG_BEST, FLG_SEQBYSEQ, FLG_NOSHRTINFO, FLG_SENSITIVE = 0x02, 0x10, 0x20, 0x80     # rmap.h:53-65
FLG_RAWRESULTS = 0x10000              # oracle only: all alignments of a call, the duplicate handling left to the caller


class OrIndex(C.Structure):
    _fields_ = [("k", C.c_int), ("s", C.c_int), ("typ", C.c_int), ("nbits_key", C.c_int), ("nbits_lo", C.c_int),
                ("nkeys", C.c_uint32), ("idx", C.POINTER(C.c_uint32)), ("pos", C.POINTER(C.c_uint32)),
                ("npos", C.c_uint32), ("maxpos", C.c_uint32), ("nwords", C.c_uint32),
                ("wordidx", C.POINTER(C.c_uint32)), ("posidx", C.POINTER(C.c_uint32)),
                ("nseq", C.c_int64), ("sop", C.POINTER(C.c_uint64)), ("totlen", C.c_uint64),
                ("packed", C.POINTER(C.c_uint32)), ("names", C.c_char_p), ("namsiz", C.c_uint64)]


class OrParams(C.Structure):
    _fields_ = [("ncut", C.c_int), ("min_cover", C.c_uint32), ("min_swatscor", C.c_int),
                ("min_swatscor_below_max", C.c_int), ("min_basq", C.c_int), ("target_depth", C.c_int),
                ("max_depth", C.c_int), ("flags", C.c_uint32), ("match", C.c_int), ("mismatch", C.c_int),
                ("gap_init", C.c_int), ("gap_ext", C.c_int)]


class OrCand(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("qs", C.c_uint32), ("qe", C.c_uint32), ("rs", C.c_uint64), ("re", C.c_uint64),
                ("band_l", C.c_int), ("band_r", C.c_int), ("sqidx", C.c_int64), ("dqo", C.c_uint32), ("dro", C.c_int),
                ("swscor", C.c_int), ("cover", C.c_uint32), ("used_simd", C.c_int)]


class OrResult(C.Structure):
    _fields_ = [("reverse", C.c_int), ("swatscor", C.c_int), ("q_start", C.c_uint32), ("q_end", C.c_uint32),
                ("s_start", C.c_uint64), ("s_end", C.c_uint64), ("sidx", C.c_int64), ("stroffs", C.c_int),
                ("strlen", C.c_int), ("cand_first", C.c_int)]


class OrAli(C.Structure):
    _fields_ = [("score", C.c_int), ("qs", C.c_int), ("qe", C.c_int), ("rs", C.c_int), ("re", C.c_int),
                ("dlen", C.c_int), ("diffstr", C.POINTER(C.c_uint8))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ODIR, "liboracle.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-s", "oracle"], cwd=ODIR, check=True)
        L = C.CDLL(so)
        L.or_index_build.restype = C.POINTER(OrIndex)
        L.or_index_build.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.c_int, C.c_int]
        L.or_index_read.restype = C.POINTER(OrIndex)
        L.or_index_read.argtypes = [C.c_char_p]
        L.or_index_write.argtypes = [C.POINTER(OrIndex), C.c_char_p]
        L.or_index_free.argtypes = [C.POINTER(OrIndex)]
        L.or_params_default.argtypes = [C.POINTER(OrParams), C.POINTER(OrIndex)]
        L.or_map_create.restype = C.c_void_p
        L.or_map_create.argtypes = [C.POINTER(OrIndex)]
        L.or_map_free.argtypes = [C.c_void_p]
        L.or_map_single.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint32, C.POINTER(OrParams)]
        L.or_map_single_restricted.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint32, C.POINTER(OrParams), C.c_int,
                                               C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.or_map_set_prevmax.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.or_map_set_seed_range.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.or_map_hit_total.restype = C.c_uint32
        L.or_map_hit_total.argtypes = [C.c_void_p, C.c_int]
        L.or_map_dump_str.restype = C.c_long
        L.or_map_dump_str.argtypes = [C.c_void_p, C.c_ulonglong, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
        L.or_index_build_fine.restype = C.POINTER(OrIndex)
        L.or_index_build_fine.argtypes = [C.POINTER(OrIndex), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int, C.c_int]
        L.or_index_free_fine.argtypes = [C.POINTER(OrIndex)]
        L.or_map_results.restype = C.POINTER(OrResult)
        L.or_map_results.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint8))]
        L.or_map_cands.restype = C.POINTER(OrCand)
        L.or_map_cands.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.or_map_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.or_sw_full.restype = C.c_int
        L.or_sw_full.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int]
        L.or_sw_band_fast.argtypes = [C.POINTER(C.c_int), C.c_char_p, C.c_uint32, C.c_char_p, C.c_int, C.c_void_p, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.or_sw_band_full.argtypes = [C.POINTER(C.POINTER(OrAli)), C.POINTER(C.c_int), C.c_char_p, C.c_uint32, C.c_char_p,
                                      C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int]
        L.or_ali_free.argtypes = [C.POINTER(OrAli), C.c_int]
        L.or_score_matrix.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.or_sort2_u32.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.or_sort_u64.argtypes = [C.c_int, C.c_void_p]
        _lib = L
    return _lib


def build_index(seqs, names, k, s):
    """seqs: list of ASCII bytes; returns POINTER(OrIndex) (caller frees with lib().or_index_free)."""
    n = len(seqs)
    A = (C.c_char_p * n)(*seqs)
    Ls = (C.c_uint32 * n)(*[len(x) for x in seqs])
    N = (C.c_char_p * n)(*[x.encode() if isinstance(x, str) else x for x in names])
    ix = lib().or_index_build(n, A, Ls, N, k, s)
    if not ix:
        raise RuntimeError("or_index_build failed")
    return ix


def build_fine_index(ix, intervals, k=5, s=1):
    """The on-the-fly index of rmapPair's rescue round over the interval windows (rmap.c:495-517); free with
    lib().or_index_free_fine BEFORE the main index."""
    n = len(intervals)
    sx = (C.c_int64 * max(1, n))(*[iv[0] for iv in intervals])
    lo = (C.c_uint32 * max(1, n))(*[iv[1] for iv in intervals])
    hi = (C.c_uint32 * max(1, n))(*[iv[2] for iv in intervals])
    return lib().or_index_build_fine(ix, n, sx, lo, hi, k, s)


def default_params(ix):
    p = OrParams()
    lib().or_params_default(C.byref(p), ix)
    return p


class Mapper:
    def __init__(self, ix):
        self.ix = ix
        self.m = lib().or_map_create(ix)

    def close(self):
        if self.m:
            lib().or_map_free(self.m)
            self.m = None

    def map(self, bases: bytes, quals, params, intervals=None, prevmax=None, seed_range=None):
        """Returns list of dicts (raw result array of the read, reference order).  intervals: list of (sx, lo, hi) =
        one mapSingleRead call of rmapPair with seeding restricted to these windows (rmap.c:438-492); prevmax: running
        score maxima of the ResultSet the call appends to."""
        if prevmax is not None:
            lib().or_map_set_prevmax(self.m, prevmax[0], prevmax[1])
        if seed_range is not None:              # (first, last) base the k-mer words come from: the second call of a split read
            lib().or_map_set_seed_range(self.m, seed_range[0], seed_range[1])
        if intervals is None:
            rv = lib().or_map_single(self.m, bases, quals, len(bases), C.byref(params))
        else:
            n_iv = len(intervals)
            sx = (C.c_int64 * max(1, n_iv))(*[iv[0] for iv in intervals])
            lo = (C.c_uint32 * max(1, n_iv))(*[iv[1] for iv in intervals])
            hi = (C.c_uint32 * max(1, n_iv))(*[iv[2] for iv in intervals])
            rv = lib().or_map_single_restricted(self.m, bases, quals, len(bases), C.byref(params), n_iv, sx, lo, hi)
        n = C.c_int()
        dp = C.POINTER(C.c_uint8)()
        rp = lib().or_map_results(self.m, C.byref(n), C.byref(dp))
        out = []
        for i in range(n.value):
            r = rp[i]
            out.append(dict(reverse=r.reverse, score=r.swatscor, q_start=r.q_start, q_end=r.q_end, s_start=r.s_start,
                            s_end=r.s_end, sidx=r.sidx, diffstr=bytes(dp[r.stroffs:r.stroffs + r.strlen])))
        self.cand_first = [bool(rp[i].cand_first) for i in range(n.value)]
        return rv, out

    def hit_total(self, ktuple_maxhit):
        return int(lib().or_map_hit_total(self.m, ktuple_maxhit))

    def dump(self, readno, name, with_hitlists=False):
        n = lib().or_map_dump_str(self.m, readno, name.encode(), int(with_hitlists), None, 0)
        buf = C.create_string_buffer(n + 1)
        lib().or_map_dump_str(self.m, readno, name.encode(), int(with_hitlists), buf, n + 1)
        return buf.value.decode()

    def stats(self):
        a = (C.c_int * 8)()
        lib().or_map_stats(self.m, a)
        return list(a)


def diffstr_to_cigar(d: bytes, q_start: int, q_end: int, qlen: int, reverse: int, ext: bool = False):
    """Render a DiffStr (diffstr.h:28-76) plus clipping as a SAM CIGAR and edit distance.
    Mirrors what the reference's SAM writer prints for the KAT reads (test/bam_cigar_test.py)."""
    ops = []  # (op, n)

    def push(op, n):
        if n <= 0:
            return
        if ops and ops[-1][0] == op:
            ops[-1][1] += n
        else:
            ops.append([op, n])
    nm = 0
    i = 0
    while i < len(d) and d[i] != 0:
        typ, cnt = d[i] >> 6, d[i] & 0x3F
        last = (i + 1 >= len(d)) or d[i + 1] == 0
        if typ == 0:      # M: cnt+1 matches
            push("M", cnt + 1)
        elif typ == 3:    # S: cnt matches then a mismatch (none at the very end)
            push("M", cnt)
            if not last:
                push("X" if ext else "M", 1)
                nm += 1
        elif typ == 2:    # I
            push("M", cnt)
            push("I", 1)
            nm += 1
        else:             # D
            push("M", cnt)
            push("D", 1)
            nm += 1
        i += 1
    left, right = q_start - 1, qlen - q_end
    if reverse:
        left, right = right, left
    s = ""
    if left:
        s += "%dS" % left
    s += "".join("%d%s" % (n, op) for op, n in ops)
    if right:
        s += "%dS" % right
    return s, nm
