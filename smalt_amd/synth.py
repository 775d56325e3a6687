"""Seeded synthetic inputs for the seed-and-extend path.

Reference: NCHR sequences of equal length, i.i.d. uniform ACGT, optionally with a fraction
of bases overwritten by diverged copies of a few repeat families so that candidate counts
are genome-like.  Reads: uniform start, 50 % reverse-complement, per-base substitutions and a
small fraction of reads carrying one 1-bp indel; constant base quality.

Everything is numpy on 2-bit codes (A0 C1 G2 T3); nothing here touches the reference
implementation or its tools.
"""
from __future__ import annotations

import numpy as np

ALPHABET = np.frombuffer(b"ACGT", dtype=np.uint8)
DEFAULT_SEED = 20261004


def make_reference(nchr: int, chrlen: int, seed: int = DEFAULT_SEED, repeat_frac: float = 0.15,
                   n_fam: int = 50, cons_len: int = 300, divergence: float = 0.08):
    """Return a list of `nchr` uint8 code arrays (values 0..3) of length `chrlen`."""
    rng = np.random.default_rng(seed)
    chroms = [rng.integers(0, 4, size=chrlen, dtype=np.uint8) for _ in range(nchr)]
    if repeat_frac > 0 and chrlen > 4 * cons_len:
        fams = rng.integers(0, 4, size=(n_fam, cons_len), dtype=np.uint8)
        ncopies = int(repeat_frac * chrlen / cons_len)
        for c in chroms:
            starts = rng.integers(0, chrlen - cons_len, size=ncopies)
            fam = rng.integers(0, n_fam, size=ncopies)
            copies = fams[fam]                                   # (ncopies, cons_len)
            mut = rng.random(copies.shape) < divergence
            copies = np.where(mut, (copies + rng.integers(1, 4, size=copies.shape)) & 3, copies)
            idx = starts[:, None] + np.arange(cons_len)[None, :]
            c[idx.ravel()] = copies.ravel().astype(np.uint8)
    return chroms


def revcomp_codes(codes: np.ndarray) -> np.ndarray:
    return (3 - codes[::-1]).astype(np.uint8)


def make_reads(chroms, n: int, length: int, seed: int = DEFAULT_SEED + 1, sub_rate: float = 0.01,
               indel_read_frac: float = 0.02):
    """Return (list of uint8 code arrays, truth array[n,3] = (chr, pos, strand))."""
    rng = np.random.default_rng(seed)
    nchr = len(chroms)
    reads = []
    truth = np.zeros((n, 3), dtype=np.int64)
    chr_ix = rng.integers(0, nchr, size=n)
    for i in range(n):
        c = chroms[chr_ix[i]]
        pos = int(rng.integers(0, len(c) - length - 2))
        r = c[pos:pos + length + 1].copy()
        if rng.random() < indel_read_frac:
            p = int(rng.integers(10, length - 10))
            if rng.random() < 0.5:                      # deletion in read
                r = np.concatenate([r[:p], r[p + 1:]])
            else:                                       # insertion in read
                r = np.concatenate([r[:p], rng.integers(0, 4, size=1, dtype=np.uint8), r[p:]])
        r = r[:length]
        mut = rng.random(length) < sub_rate
        r = np.where(mut, (r + rng.integers(1, 4, size=length)) & 3, r).astype(np.uint8)
        strand = int(rng.random() < 0.5)
        if strand:
            r = revcomp_codes(r)
        reads.append(r)
        truth[i] = (chr_ix[i], pos, strand)
    return reads, truth


def make_long_reads(chroms, n: int, length: int, seed: int = DEFAULT_SEED + 2, sub: float = 0.03,
                    ins: float = 0.05, dele: float = 0.04):
    """PacBio-shape reads: per-base sub/ins/del rates over a `length` bp source."""
    rng = np.random.default_rng(seed)
    reads = []
    truth = np.zeros((n, 3), dtype=np.int64)
    for i in range(n):
        ci = int(rng.integers(0, len(chroms)))
        c = chroms[ci]
        pos = int(rng.integers(0, len(c) - length - 1))
        src = c[pos:pos + length]
        u = rng.random(length)
        out = []
        keep = u >= dele
        subm = (u >= dele) & (u < dele + sub)
        base = np.where(subm, (src + rng.integers(1, 4, size=length)) & 3, src).astype(np.uint8)
        insm = rng.random(length) < ins
        insb = rng.integers(0, 4, size=length, dtype=np.uint8)
        for j in range(length):
            if keep[j]:
                out.append(base[j])
            if insm[j]:
                out.append(insb[j])
        r = np.array(out, dtype=np.uint8)
        strand = int(rng.random() < 0.5)
        if strand:
            r = revcomp_codes(r)
        reads.append(r)
        truth[i] = (ci, pos, strand)
    return reads, truth


def codes_to_ascii(codes: np.ndarray) -> bytes:
    return ALPHABET[codes].tobytes()


def write_fasta(path: str, chroms, prefix: str = "chr", width: int = 60) -> None:
    with open(path, "wb") as f:
        for i, c in enumerate(chroms):
            f.write(b">%s%d\n" % (prefix.encode(), i + 1))
            s = codes_to_ascii(c)
            for o in range(0, len(s), width):
                f.write(s[o:o + width] + b"\n")


def write_fastq(path: str, reads, truth=None, qual: bytes = b"I", prefix: str = "r") -> None:
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            name = b"%s%d" % (prefix.encode(), i)
            if truth is not None:
                name += b"_%d_%d_%d" % tuple(int(x) for x in truth[i])
            s = r if isinstance(r, (bytes, bytearray)) else codes_to_ascii(r)
            f.write(b"@" + name + b"\n" + s + b"\n+\n" + qual * len(s) + b"\n")


def make_pairs(chroms, n: int, length: int, seed: int = DEFAULT_SEED + 3, insert_mean: float = 300.0, insert_sd: float = 30.0,
               sub_rate: float = 0.01, indel_read_frac: float = 0.02):
    """Paired-end reads, FR orientation (BASELINE configs[2]): a fragment of length N(insert_mean,
    insert_sd) is drawn from a random locus and strand; read 1 is its first `length` bases, read 2 the reverse complement
    of its last `length` bases.  Returns (reads1, reads2, truth[n,4] = (chr, fragment start, strand, fragment length))."""
    rng = np.random.default_rng(seed)
    r1, r2 = [], []
    truth = np.zeros((n, 4), dtype=np.int64)

    def noisy(r):
        r = r.copy()
        if rng.random() < indel_read_frac and len(r) > 24:
            p = int(rng.integers(10, len(r) - 10))
            if rng.random() < 0.5:
                r = np.concatenate([r[:p], r[p + 1:], rng.integers(0, 4, size=1, dtype=np.uint8)])
            else:
                r = np.concatenate([r[:p], rng.integers(0, 4, size=1, dtype=np.uint8), r[p:-1]])
        mut = rng.random(len(r)) < sub_rate
        return np.where(mut, (r + rng.integers(1, 4, size=len(r))) & 3, r).astype(np.uint8)

    for i in range(n):
        ci = int(rng.integers(0, len(chroms)))
        c = chroms[ci]
        flen = max(length, int(round(rng.normal(insert_mean, insert_sd))))
        pos = int(rng.integers(0, len(c) - flen - 1))
        frag = c[pos:pos + flen]
        strand = int(rng.random() < 0.5)
        if strand:
            frag = revcomp_codes(frag)
        r1.append(noisy(frag[:length]))
        r2.append(noisy(revcomp_codes(frag[-length:])))
        truth[i] = (ci, pos, strand, flen)
    return r1, r2, truth
