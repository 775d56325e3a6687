"""Writer of SMALT's index files from plain arrays, so that an index built in HBM can be handed
to the unmodified CPU reference (bench.py's cpu_baseline leg) or kept for later runs.

  <prefix>.sma  reference set  (seqSetWriteBinFil, sequence.c:2448-2519)
  <prefix>.smi  hash index     (hashTableWrite,   hashidx.c:1214-1255)
both inside the 12-word container of filio.c:48-68; little-endian uint32 throughout.
"""
from __future__ import annotations

import numpy as np

FILIO_SIG, FILIO_ENDIAN = 0x73212173, 0x6E378A19


def _container(f, siz_words: int, typ: int, version: int, header) -> None:
    h = np.zeros(12, dtype=np.uint32)
    h[0], h[1], h[2], h[3], h[4], h[5] = FILIO_SIG, FILIO_ENDIAN, (siz_words + 12) & 0xFFFFFFFF, typ, version, len(header)
    h.tofile(f)
    np.asarray(header, dtype=np.uint32).tofile(f)


def write_sma(prefix: str, names, sop, packed: np.ndarray) -> None:
    nseq = len(names)
    namebytes = b"".join(n.encode() + b"\0" for n in names)
    tot = int(sop[-1])
    hdr = [nseq & 0xFFFFFFFF, nseq >> 32, len(namebytes) & 0xFFFFFFFF, len(namebytes) >> 32, tot & 0xFFFFFFFF, tot >> 32, 2, 0]
    seqsiz = tot // 10 + 1
    totsiz = 8 + seqsiz + nseq + ((len(namebytes) - 1) // 4 + 1)
    with open(prefix + ".sma", "wb") as f:
        _container(f, totsiz, 1, 4, hdr)
        f.write(namebytes)
        np.diff(np.asarray(sop, dtype=np.uint64)).astype(np.uint32).tofile(f)
        np.ascontiguousarray(packed[:seqsiz]).view(np.uint32).tofile(f)


def write_smi_perfect(prefix: str, k: int, s: int, idx: np.ndarray, pos: np.ndarray, maxpos: int) -> None:
    npos = int(pos.shape[0])
    hdr = [k, s, npos, maxpos, 0, 2 * k, 0, 0]
    with open(prefix + ".smi", "wb") as f:
        _container(f, npos + idx.shape[0], 2, 3, hdr)
        np.ascontiguousarray(idx).view(np.uint32).tofile(f)
        np.ascontiguousarray(pos).view(np.uint32).tofile(f)
