"""ctypes binding of libsmaltgpu.so (C ABI in include/smaltgpu.h) and the host-side mirror of the
reference's per-read mapping interface (rmap.h:83-145): `Index` stands for the (HashTable,
SeqSet) pair that `rmapCreate` receives, `Mapper` for the `RMap`, `Mapper.map_batch` for a
block of `rmapSingle` calls followed by `rmapGetData`.

There is no CPU fallback: if the HIP library is missing or no device is present every call
raises `SmaltGpuError`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(HERE, "libsmaltgpu.so")

FLG_BEST, FLG_SEQBYSEQ, FLG_NOSHRTINFO, FLG_SENSITIVE = 0x02, 0x10, 0x20, 0x80


class SmaltGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("smaltgpu error %d: %s" % (code, msg))
        self.code = code


class IndexDesc(C.Structure):
    _fields_ = [("k", C.c_int32), ("s", C.c_int32), ("typ", C.c_int32), ("nbits_key", C.c_int32), ("nbits_lo", C.c_int32),
                ("npos", C.c_uint32), ("nwords", C.c_uint32), ("idx", C.c_void_p), ("pos", C.c_void_p),
                ("wordidx", C.c_void_p), ("posidx", C.c_void_p), ("nseq", C.c_int64), ("sop", C.c_void_p),
                ("packed", C.c_void_p), ("on_device", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("ktuple_maxhit", C.c_int32), ("min_cover", C.c_uint32), ("min_swatscor", C.c_int32),
                ("min_swatscor_below_max", C.c_int32), ("min_basqval", C.c_int32), ("target_depth", C.c_int32),
                ("max_depth", C.c_int32), ("rmapflg", C.c_uint32), ("match", C.c_int32), ("mismatch", C.c_int32),
                ("gap_init", C.c_int32), ("gap_ext", C.c_int32), ("min_cover_frac", C.c_double)]


class Result(C.Structure):
    _fields_ = [("swatscor", C.c_int32), ("q_start", C.c_uint32), ("q_end", C.c_uint32), ("s_start", C.c_uint64),
                ("s_end", C.c_uint64), ("sidx", C.c_int32), ("reverse", C.c_uint32), ("stroffs", C.c_uint32),
                ("strlen", C.c_uint32)]


class ReadStat(C.Structure):
    _fields_ = [("swatscor_max", C.c_int32), ("swatscor_2ndmax", C.c_int32), ("n_ali_done", C.c_int32),
                ("n_ali_tot", C.c_int32), ("n_hits_used", C.c_uint32), ("n_hits_tot", C.c_uint32),
                ("errcode", C.c_int32), ("nres", C.c_uint32), ("max1scor", C.c_int32), ("errsite", C.c_int32)]


class Interval(C.Structure):
    _fields_ = [("sidx", C.c_int32), ("lo", C.c_uint32), ("hi", C.c_uint32)]


class CallCtx(C.Structure):
    _fields_ = [("iv_off", C.POINTER(C.c_uint64)), ("iv", C.POINTER(Interval)), ("min_swatscor", C.POINTER(C.c_int32)),
                ("prev_max", C.POINTER(C.c_int32)), ("fine_index", C.c_int32), ("raw_alignments", C.c_int32), ("hitlist_len", C.POINTER(C.c_uint32)), ("seed_range", C.POINTER(C.c_uint32))]


class PostResult(C.Structure):
    _fields_ = [("swatscor", C.c_int32), ("q_start", C.c_uint32), ("q_end", C.c_uint32), ("s_start", C.c_uint64), ("s_end", C.c_uint64),
                ("sidx", C.c_int32), ("status", C.c_uint32), ("mapscor", C.c_int32), ("prob", C.c_double), ("rsltx", C.c_int16),
                ("qsegx", C.c_int16), ("swrank", C.c_int16), ("pad", C.c_int16), ("stroffs", C.c_uint32), ("strlen", C.c_uint32)]


class PostOut(C.Structure):
    _fields_ = [("nreads", C.c_uint32), ("res_off", C.POINTER(C.c_uint64)), ("res", C.POINTER(PostResult)), ("diffstr", C.POINTER(C.c_uint8)),
                ("sort_off", C.POINTER(C.c_uint64)), ("sortr", C.POINTER(C.c_int32)), ("segsrtr", C.POINTER(C.c_int32)),
                ("seg_off", C.POINTER(C.c_uint64)), ("segnor", C.POINTER(C.c_int32)), ("qsegno", C.POINTER(C.c_int32)),
                ("setstatus", C.POINTER(C.c_uint32)), ("needs_reference", C.POINTER(C.c_int32))]


class ReadsView(C.Structure):          # smaltgpu_reads_view
    _fields_ = [("nreads", C.c_uint32), ("has_qual", C.c_uint32), ("bases", C.POINTER(C.c_uint8)), ("quals", C.POINTER(C.c_uint8)),
                ("read_off", C.POINTER(C.c_uint64)), ("names", C.POINTER(C.c_char)), ("name_off", C.POINTER(C.c_uint64)), ("consumed", C.c_uint64)]


class ReportOpts(C.Structure):         # smaltgpu_report_opts
    _fields_ = [("format", C.c_int32), ("modflags", C.c_uint32), ("outflags", C.c_uint32), ("min_swscor", C.c_int32),
                ("min_swscor_below_max", C.c_int32), ("min_identity", C.c_double)]


FMT_CIGAR, FMT_SAM, FMT_SSAHA = 0, 1, 2
REP_SOFTCLIP, REP_HEADER, REP_XMISMATCH = 0x02, 0x04, 0x08
OUT_BEST, OUT_SINGLE, OUT_SPLIT, OUT_RANDSEL = 0x01, 0x02, 0x04, 0x08


class PairOpts(C.Structure):           # smaltgpu_pair_opts
    _fields_ = [("insert_min", C.c_int32), ("insert_max", C.c_int32), ("library", C.c_int32), ("every_pair", C.c_int32), ("nthreads", C.c_int32)]


class PairInfo(C.Structure):           # smaltgpu_pair_info
    _fields_ = [("pairflg", C.c_uint8), ("rounds", C.c_uint8), ("nali", C.c_uint16 * 2)]


LIB_PE, LIB_MP, LIB_PP, LIB_ANY = 1, 2, 3, 4


class ResidentReads(C.Structure):      # smaltgpu_resident_reads
    _fields_ = [("d_bases", C.c_void_p * 2), ("d_quals", C.c_void_p * 2), ("d_read_off", C.c_void_p * 2), ("read_off", C.c_void_p * 2), ("nreads", C.c_uint32 * 2)]


class MapperOpts(C.Structure):
    _fields_ = [("cands_per_read", C.c_uint32), ("slot_budget_gb", C.c_uint32)]


class BatchOut(C.Structure):
    _fields_ = [("nreads", C.c_uint32), ("res_off", C.POINTER(C.c_uint64)), ("res", C.POINTER(Result)),
                ("diffstr", C.POINTER(C.c_uint8)), ("stat", C.POINTER(ReadStat))]


_lib = None


def lib():
    """Load libsmaltgpu.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise SmaltGpuError(-1, "libsmaltgpu.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        try:                       # torch bundles its own HIP runtime: load it first so that one runtime serves both
            import torch  # noqa: F401
        except Exception:          # torch is plumbing for bench/tests, not a dependency of the library
            pass
        L = C.CDLL(LIBPATH)
        L.smaltgpu_last_error.restype = C.c_char_p
        L.smaltgpu_timer_name.restype = C.c_char_p
        L.smaltgpu_index_load.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
        L.smaltgpu_index_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(IndexDesc), C.c_int]
        L.smaltgpu_index_free.argtypes = [C.c_void_p]
        L.smaltgpu_index_build.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_char_p), C.c_int64,
                                           C.c_int32, C.c_int32, C.POINTER(C.c_float)]
        L.smaltgpu_index_build_device.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_char_p), C.c_int64,
                                                  C.c_int32, C.c_int32, C.POINTER(C.c_float)]
        L.smaltgpu_index_save.argtypes = [C.c_void_p, C.c_char_p]
        L.smaltgpu_index_info.argtypes = [C.c_void_p, C.POINTER(IndexDesc)]
        L.smaltgpu_params_default.argtypes = [C.POINTER(Params), C.c_void_p]
        L.smaltgpu_mapper_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint32, C.c_uint32]
        L.smaltgpu_mapper_create_ex.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(MapperOpts)]
        L.smaltgpu_index_clone.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int]
        L.smaltgpu_mapper_free.argtypes = [C.c_void_p]
        L.smaltgpu_map_batch.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_uint32,
                                         C.POINTER(Params), C.POINTER(BatchOut)]
        L.smaltgpu_map_batch_ctx.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_uint32,
                                             C.POINTER(Params), C.POINTER(CallCtx), C.POINTER(BatchOut)]
        L.smaltgpu_hit_totals.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(Params), C.POINTER(C.c_uint32)]
        L.smaltgpu_post_create.restype = C.c_void_p
        L.smaltgpu_post_free.argtypes = [C.c_void_p]
        L.smaltgpu_postprocess.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64, C.POINTER(BatchOut), C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64),
                                           C.c_void_p, C.POINTER(Params), C.c_int, C.POINTER(PostOut)]
        L.smaltgpu_reads_create.restype = C.c_void_p
        L.smaltgpu_reads_free.argtypes = [C.c_void_p]
        L.smaltgpu_reads_parse.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int, C.c_uint32, C.c_int, C.POINTER(ReadsView)]
        L.smaltgpu_map_split.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(Params), C.c_void_p, C.c_int,
                                         C.POINTER(PostOut), C.POINTER(C.c_uint32)]
        L.smaltgpu_report_create.restype = C.c_void_p
        L.smaltgpu_report_free.argtypes = [C.c_void_p]
        L.smaltgpu_report_header.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.c_int64, C.POINTER(ReportOpts), C.c_char_p, C.c_char_p,
                                             C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.smaltgpu_report_emit.argtypes = [C.c_void_p, C.POINTER(PostOut), C.POINTER(BatchOut), C.POINTER(ReadsView), C.POINTER(C.c_char_p), C.c_int64,
                                           C.POINTER(ReportOpts), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.smaltgpu_pairs_create.restype = C.c_void_p
        L.smaltgpu_pairs_free.argtypes = [C.c_void_p]
        L.smaltgpu_map_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Params),
                                         C.POINTER(PairOpts), C.c_void_p]
        L.smaltgpu_pairs_timers.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.smaltgpu_map_pairs_resident.argtypes = [C.c_void_p, C.POINTER(ResidentReads), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Params),
                                                  C.POINTER(PairOpts), C.c_void_p]
        L.smaltgpu_pairs_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.POINTER(PairInfo)), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.smaltgpu_report_emit_pairs.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ReadsView), C.POINTER(ReadsView), C.POINTER(C.c_char_p), C.c_int64,
                                                 C.POINTER(ReportOpts), C.POINTER(PairOpts), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.smaltgpu_index_seqnames.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_int64)]
        L.smaltgpu_map_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                                C.POINTER(Params)]
        L.smaltgpu_fetch_results.argtypes = [C.c_void_p, C.POINTER(BatchOut)]
        L.smaltgpu_synchronize.argtypes = [C.c_void_p]
        L.smaltgpu_timers.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
        L.smaltgpu_set_debug.argtypes = [C.c_void_p, C.c_int]
        L.smaltgpu_dump_read.restype = C.c_long
        L.smaltgpu_dump_read.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_size_t]
        L.smaltgpu_sw_full_batch.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32), C.c_char_p,
                                             C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(Params), C.POINTER(C.c_int32), C.c_int]
        L.smaltgpu_rank_sort_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _check(rv):
    if rv != 0:
        raise SmaltGpuError(rv, lib().smaltgpu_last_error().decode())


def device_count() -> int:
    return lib().smaltgpu_device_count()


class Index:
    """Index image resident in HBM (replaces hashTableRead + seqSetReadBinFil)."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def load(cls, prefix: str, device: int = 0) -> "Index":
        h = C.c_void_p()
        _check(lib().smaltgpu_index_load(C.byref(h), prefix.encode(), device))
        return cls(h)

    @classmethod
    def from_desc(cls, desc: IndexDesc, device: int = 0) -> "Index":
        h = C.c_void_p()
        _check(lib().smaltgpu_index_create(C.byref(h), C.byref(desc), device))
        return cls(h)

    @classmethod
    def build(cls, seqs: Sequence[bytes], names: Sequence[str], k: int, s: int, device: int = 0) -> "Index":
        """Build the index image on the device from reference sequences (replaces `smalt index`: hashTableSetUp
        hashidx.c:829-998 + seqSetCompress sequence.c:1360-1424).  `build_ms` holds the device time."""
        cat = b"".join(seqs)
        off = (C.c_uint64 * (len(seqs) + 1))()
        o = 0
        for i, q in enumerate(seqs):
            off[i] = o
            o += len(q)
        off[len(seqs)] = o
        nm = (C.c_char_p * len(seqs))(*[n.encode() for n in names])
        h = C.c_void_p()
        ms = C.c_float(0.0)
        _check(lib().smaltgpu_index_build(C.byref(h), device, cat, off, nm, C.c_int64(len(seqs)), k, s, C.byref(ms)))
        ix = cls(h)
        ix.build_ms = float(ms.value)
        return ix

    @classmethod
    def build_device(cls, d_bases_ptr: int, seq_off: Sequence[int], names: Sequence[str], k: int, s: int, device: int = 0) -> "Index":
        """Same from bases already in HBM (`d_bases_ptr`: device address of the concatenated sequences)."""
        n = len(names)
        off = (C.c_uint64 * (n + 1))(*[int(x) for x in seq_off])
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        h = C.c_void_p()
        ms = C.c_float(0.0)
        _check(lib().smaltgpu_index_build_device(C.byref(h), device, C.c_void_p(d_bases_ptr), off, nm, C.c_int64(n), k, s, C.byref(ms)))
        ix = cls(h)
        ix.build_ms = float(ms.value)
        return ix

    def clone(self, device: int) -> "Index":
        """A second image on `device`, copied device to device (xGMI) from this one."""
        h = C.c_void_p()
        _check(lib().smaltgpu_index_clone(C.byref(h), self.h, device))
        return Index(h)

    def save(self, prefix: str) -> None:
        """Write <prefix>.sma / <prefix>.smi, byte-compatible with the reference's index files."""
        _check(lib().smaltgpu_index_save(self.h, prefix.encode()))

    def info(self) -> IndexDesc:
        d = IndexDesc()
        _check(lib().smaltgpu_index_info(self.h, C.byref(d)))
        return d

    def default_params(self) -> Params:
        p = Params()
        lib().smaltgpu_params_default(C.byref(p), self.h)
        return p

    def close(self):
        if self.h:
            lib().smaltgpu_index_free(self.h)
            self.h = None


class Mapper:
    """Work buffers + stream for one host thread (the analogue of an RMap, rmap.h:83)."""

    def __init__(self, index: Index, max_batch_reads: int, max_read_len: int, cands_per_read: int = 0, slot_budget_gb: int = 0):
        self.index = index
        self.h = C.c_void_p()
        opts = MapperOpts(cands_per_read, slot_budget_gb)
        _check(lib().smaltgpu_mapper_create_ex(C.byref(self.h), index.h, max_batch_reads, max_read_len, C.byref(opts)))

    def close(self):
        if self.h:
            lib().smaltgpu_mapper_free(self.h)
            self.h = None

    def set_history(self, on: bool = True):
        """Serial-order mode (smaltgpu_mapper_set_history): consecutive map_batch calls form one serial run of `smalt map -n 0`
        as far as the hit-list capacity goes; calling it again starts a new run."""
        _check(lib().smaltgpu_mapper_set_history(self.h, 1 if on else 0))

    def set_debug(self, level: int):
        _check(lib().smaltgpu_set_debug(self.h, level))

    @staticmethod
    def _pack(reads: Sequence[bytes], quals: Optional[Sequence[bytes]]):
        n = len(reads)
        off = (C.c_uint64 * (n + 1))()
        t = 0
        for i, r in enumerate(reads):
            off[i] = t
            t += len(r)
        off[n] = t
        return b"".join(reads), (b"".join(quals) if quals is not None else None), off

    def map_split(self, reads: Sequence[bytes], quals: Optional[Sequence[bytes]], params: Params, post, nthreads: int = 2):
        """smaltgpu_map_split: both calls of every read (smalt map -p) and the post-call passes.  post: a handle of
        lib().smaltgpu_post_create() that owns the arrays -> (PostOut, number of reads that got a second call)"""
        bases, q, off = self._pack(reads, quals)
        out = PostOut()
        nsec = C.c_uint32()
        _check(lib().smaltgpu_map_split(self.h, post, bases, q, off, len(reads), C.byref(params), self.index.h, nthreads, C.byref(out), C.byref(nsec)))
        return out, nsec.value

    def map_batch(self, reads: Sequence[bytes], quals: Optional[Sequence[bytes]], params: Params, allow_read_errors: bool = False):
        """-> (list per read of result dicts in the reference's raw order, list of stat dicts).
        allow_read_errors: do not raise when single reads failed (device-side limit or the reference's own per-read
        errors such as ERRCODE_SWATSCOR); their stat carries `err` and they have no results."""
        bases, q, off = self._pack(reads, quals)
        out = BatchOut()
        rv = lib().smaltgpu_map_batch(self.h, bases, q, off, len(reads), C.byref(params), C.byref(out))
        if rv != 0 and not (allow_read_errors and rv in (-5, -6, -8) and out.nreads == len(reads)):
            _check(rv)
        return self._unpack(out)

    def map_batch_ctx(self, reads: Sequence[bytes], quals: Optional[Sequence[bytes]], params: Params, intervals=None, min_swatscor=None,
                      prev_max=None, fine_index: bool = False, raw_alignments: bool = False, seed_range=None):
        """One round of rmapPair over a batch (smaltgpu_map_batch_ctx): intervals = per read a list of (sidx, lo, hi) or None
        for no restriction at all; min_swatscor = per-read thresholds; prev_max = per read (max, 2ndmax) of the ResultSet
        the call appends to; fine_index = seed against the on-the-fly k=5 index of the intervals; seed_range = per read the
        (first, last) base its k-mer words come from (the second call of a split read).
        -> (results, stats, cand_first flags per result)"""
        bases, q, off = self._pack(reads, quals)
        n = len(reads)
        ctx = CallCtx()
        keep = []
        if intervals is not None:
            ivo = (C.c_uint64 * (n + 1))()
            flat = [iv for lst in intervals for iv in lst]
            t = 0
            for i, lst in enumerate(intervals):
                ivo[i] = t
                t += len(lst)
            ivo[n] = t
            arr = (Interval * max(1, len(flat)))(*[Interval(a, b, c) for a, b, c in flat])
            ctx.iv_off, ctx.iv = ivo, arr
            keep += [ivo, arr]
        if min_swatscor is not None:
            ms = (C.c_int32 * max(1, n))(*min_swatscor)
            ctx.min_swatscor = ms
            keep.append(ms)
        if prev_max is not None:
            pm = (C.c_int32 * max(1, 2 * n))(*[x for pr_ in prev_max for x in pr_])
            ctx.prev_max = pm
            keep.append(pm)
        if seed_range is not None:
            sr = (C.c_uint32 * max(1, 2 * n))(*[x for pr_ in seed_range for x in pr_])
            ctx.seed_range = sr
            keep.append(sr)
        ctx.fine_index = 1 if fine_index else 0
        ctx.raw_alignments = 1 if raw_alignments else 0
        out = BatchOut()
        _check(lib().smaltgpu_map_batch_ctx(self.h, bases, q, off, n, C.byref(params), C.byref(ctx), C.byref(out)))
        res, stats = self._unpack(out)
        cf = [[bool(out.res[j].reverse & 2) for j in range(out.res_off[i], out.res_off[i + 1])] for i in range(n)]
        return res, stats, cf

    def hit_totals(self, reads: Sequence[bytes], quals: Optional[Sequence[bytes]], params: Params) -> List[int]:
        """calcTotalNumberOfHits (rmap.c:1076) per read: what rmapPair compares to pick the mate it maps first."""
        bases, q, off = self._pack(reads, quals)
        out = (C.c_uint32 * max(1, len(reads)))()
        _check(lib().smaltgpu_hit_totals(self.h, bases, q, off, len(reads), C.byref(params), out))
        return list(out)[:len(reads)]

    def map_pairs_raw(self, bases1, off1, quals1, bases2, off2, quals2, params: Params, popts: PairOpts, pairs_handle=None):
        """smaltgpu_map_pairs on contiguous host arrays (numpy uint8 bases / uint64 offsets per mate file): rmapPair for a block.
        -> (handle of the mapped block, numpy view of smaltgpu_pair_info, calls per round, host ms per round); the handle goes to
        smaltgpu_report_emit_pairs and is freed with lib().smaltgpu_pairs_free (or handed back in as pairs_handle)."""
        import numpy as np
        L = lib()
        h = pairs_handle or C.c_void_p(L.smaltgpu_pairs_create())
        n = len(off1) - 1
        ptr = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
        _check(L.smaltgpu_map_pairs(self.h, ptr(bases1), ptr(quals1), ptr(off1), ptr(bases2), ptr(quals2), ptr(off2), n, C.byref(params), C.byref(popts), h))
        npairs = C.c_uint32()
        info = C.POINTER(PairInfo)()
        calls = (C.c_uint64 * 4)()
        ms = (C.c_double * 4)()
        _check(L.smaltgpu_pairs_info(h, C.byref(npairs), C.byref(info), calls, ms))
        arr = np.ctypeslib.as_array(C.cast(info, C.POINTER(C.c_uint8)), shape=(max(1, npairs.value), 6))[:npairs.value]
        return h, arr, list(calls), list(ms)

    def map_batch_raw(self, bases, off, quals, params: Params) -> BatchOut:
        """smaltgpu_map_batch on contiguous host arrays (numpy uint8 bases / uint64 offsets); the returned
        views stay valid until the next call on this mapper."""
        out = BatchOut()
        _check(lib().smaltgpu_map_batch(self.h, C.cast(bases.ctypes.data, C.c_char_p), None if quals is None else C.cast(quals.ctypes.data, C.c_char_p),
                                        C.cast(off.ctypes.data, C.POINTER(C.c_uint64)), len(off) - 1, C.byref(params), C.byref(out)))
        return out

    @staticmethod
    def _unpack(out: BatchOut):
        res, stats = [], []
        for i in range(out.nreads):
            rr = []
            for j in range(out.res_off[i], out.res_off[i + 1]):
                r = out.res[j]
                rr.append(dict(reverse=int(r.reverse & 1), score=r.swatscor, q_start=r.q_start, q_end=r.q_end, s_start=r.s_start,
                               s_end=r.s_end, sidx=r.sidx, diffstr=bytes(out.diffstr[r.stroffs:r.stroffs + r.strlen])))
            res.append(rr)
            s = out.stat[i]
            stats.append(dict(swmax=s.swatscor_max, sw2nd=s.swatscor_2ndmax, nseg=s.n_ali_done, nseg_tot=s.n_ali_tot,
                              nhit=s.n_hits_used, nhit_tot=s.n_hits_tot, err=s.errcode, max1=s.max1scor, errsite=s.errsite))
        return res, stats

    def map_batch_device(self, d_bases: int, d_quals: int, d_off: int, nreads: int, total_bases: int, params: Params):
        _check(lib().smaltgpu_map_batch_device(self.h, d_bases, d_quals, d_off, nreads, total_bases, C.byref(params)))

    def synchronize(self):
        _check(lib().smaltgpu_synchronize(self.h))

    def fetch_results(self):
        out = BatchOut()
        _check(lib().smaltgpu_fetch_results(self.h, C.byref(out)))
        return out

    def fetch_begin(self):
        """Wait for the enqueued batch and start copying its results; the next map_batch_device may follow at once."""
        _check(lib().smaltgpu_fetch_begin(self.h))

    def fetch_end(self):
        out = BatchOut()
        _check(lib().smaltgpu_fetch_end(self.h, C.byref(out)))
        return out

    def timers(self):
        ms = (C.c_double * 32)()
        wk = (C.c_uint64 * 32)()
        n = lib().smaltgpu_timers(self.h, ms, wk, 32)
        names = [lib().smaltgpu_timer_name(i).decode() for i in range(n)]
        return dict(zip(names, list(ms)[:n])), list(wk)

    def dump_read(self, i: int, name: str) -> str:
        n = lib().smaltgpu_dump_read(self.h, i, name.encode(), None, 0)
        if n < 0:
            _check(int(n))
        buf = C.create_string_buffer(n + 1)
        lib().smaltgpu_dump_read(self.h, i, name.encode(), buf, n + 1)
        return buf.value.decode()

    def sw_full_batch(self, queries: Sequence[bytes], windows: Sequence[bytes], params: Params, packed16: bool = False) -> List[int]:
        """Stand-alone K2a over explicit 3-bit code arrays (bytes of codes 0..7); packed16 selects the
        two-tasks-per-lane-group 16-bit kernel (-2 for queries with non-ACGT codes)."""
        n = len(queries)
        qo = (C.c_uint32 * (n + 1))()
        ro = (C.c_uint32 * (n + 1))()
        a = b = 0
        for i in range(n):
            qo[i], ro[i] = a, b
            a += len(queries[i])
            b += len(windows[i])
        qo[n], ro[n] = a, b
        sc = (C.c_int32 * n)()
        _check(lib().smaltgpu_sw_full_batch(self.h, b"".join(queries), qo, b"".join(windows), ro, n, C.byref(params), sc, int(packed16)))
        return list(sc)

    def rank_sort_batch(self, arrays, nneed: int = -1, in_lds: bool = True):
        """Stand-alone candidate ranking sort: list of uint32 numpy arrays -> list of (keys, permutation)."""
        import numpy as np
        off = np.zeros(len(arrays) + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(a) for a in arrays])
        keys = np.ascontiguousarray(np.concatenate(arrays) if arrays else np.zeros(0), dtype=np.uint32)
        ok = np.zeros(max(1, keys.size), dtype=np.uint32)
        oi = np.zeros(max(1, keys.size), dtype=np.uint32)
        _check(lib().smaltgpu_rank_sort_batch(self.h, keys.ctypes.data, off.ctypes.data, len(arrays), nneed, int(in_lds),
                                              ok.ctypes.data, oi.ctypes.data))
        return [(ok[off[t]:off[t + 1]], oi[off[t]:off[t + 1]]) for t in range(len(arrays))]
