"""Synthetic reference + k-mer index built directly in HBM with torch ops (setup plumbing for
bench.py and the large-size tests; not part of the timed path).

The layout produced is exactly the reference's:
  packed : 3 bits/base, 10 bases per uint32, first base in bits 29..27, terminator 7 after the
           last base (sequence.c:1360-1424)
  idx/pos: perfect index, idx[key+1]-idx[key] = number of sampled k-mers with 2-bit word `key`,
           pos = k-mer serial numbers (global base offset / s) ascending per key
           (hashidx.c:829-998); only k-mers that start on the global stride grid and lie fully
           inside one sequence are indexed (hashidx.c:465-531).
Only the perfect index type is built here (4^k <= 2 * totlen / s, smalt.c:268-332).
"""
from __future__ import annotations

import numpy as np
import torch


def make_reference_gpu(nchr: int, chrlen: int, seed: int, device, repeat_frac: float = 0.15, n_fam: int = 50,
                       cons_len: int = 300, divergence: float = 0.08) -> torch.Tensor:
    """-> uint8 tensor [nchr*chrlen] of 2-bit codes (concatenated sequences)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ref = torch.randint(0, 4, (nchr * chrlen,), dtype=torch.uint8, device=device, generator=g)
    if repeat_frac > 0:
        fams = torch.randint(0, 4, (n_fam, cons_len), dtype=torch.uint8, device=device, generator=g)
        ncopies = int(repeat_frac * chrlen / cons_len)
        ar = torch.arange(cons_len, device=device)
        for c in range(nchr):
            starts = torch.randint(0, chrlen - cons_len, (ncopies,), device=device, generator=g)
            fam = torch.randint(0, n_fam, (ncopies,), device=device, generator=g)
            copies = fams[fam]
            mut = torch.rand(copies.shape, device=device, generator=g) < divergence
            sub = torch.randint(1, 4, copies.shape, dtype=torch.uint8, device=device, generator=g)
            copies = torch.where(mut, (copies + sub) & 3, copies)
            idx = (starts[:, None] + ar[None, :]).reshape(-1) + c * chrlen
            ref[idx] = copies.reshape(-1)
    return ref


def pack_reference(ref: torch.Tensor) -> torch.Tensor:
    """3-bit packing: -> int32 tensor [totlen // 10 + 1]."""
    tot = ref.numel()
    nw = tot // 10 + 1
    padded = torch.full((nw * 10,), 0, dtype=torch.int32, device=ref.device)
    padded[:tot] = ref.to(torch.int32)
    padded[tot] = 7
    w = padded.view(nw, 10)
    shifts = torch.arange(27, -1, -3, device=ref.device, dtype=torch.int32)
    return (w << shifts[None, :]).sum(dim=1, dtype=torch.int32)


def build_perfect_index(ref: torch.Tensor, sop: np.ndarray, k: int, s: int):
    """-> (idx int32[4^k+1], pos int32[npos]) as device tensors (bit patterns of uint32)."""
    dev = ref.device
    tot = int(sop[-1])
    nkeys = 4 ** k
    assert nkeys <= 2 * (tot // s), "index type would be HASH32MIX (smalt.c:298); not built here"
    chunks_key, chunks_pos = [], []
    for i in range(len(sop) - 1):
        lo, hi = int(sop[i]), int(sop[i + 1])
        g0 = ((lo + s - 1) // s) * s                      # first grid position inside the sequence
        if hi - g0 < k:
            continue
        nk = (hi - k - g0) // s + 1
        p = g0 + torch.arange(nk, device=dev, dtype=torch.int64) * s
        word = torch.zeros(nk, dtype=torch.int64, device=dev)
        for j in range(k):
            word = (word << 2) | ref[p + j].to(torch.int64)
        chunks_key.append(word)
        chunks_pos.append(p // s)
        del word, p
    key = torch.cat(chunks_key)
    pos = torch.cat(chunks_pos)
    del chunks_key, chunks_pos
    comp = (key << 32) | pos                              # sort by (key, pos)
    del key, pos
    comp, _ = torch.sort(comp)
    pos_sorted = (comp & 0xFFFFFFFF).to(torch.int32)
    key_sorted = comp >> 32
    del comp
    counts = torch.bincount(key_sorted, minlength=nkeys)
    del key_sorted
    idx = torch.zeros(nkeys + 1, dtype=torch.int64, device=dev)
    idx[1:] = torch.cumsum(counts, 0)
    return idx.to(torch.int32), pos_sorted


def make_reads_gpu(ref: torch.Tensor, sop: np.ndarray, n: int, length: int, seed: int, sub_rate: float = 0.01,
                   indel_read_frac: float = 0.02):
    """-> (ASCII uint8 tensor [n*length], truth int64 [n,3] (seq, pos, strand)); all on device."""
    dev = ref.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    nseq = len(sop) - 1
    seq = torch.randint(0, nseq, (n,), device=dev, generator=g)
    sop_t = torch.as_tensor(np.asarray(sop, dtype=np.int64), device=dev)
    slen = sop_t[seq + 1] - sop_t[seq]
    pos = (torch.rand(n, device=dev, generator=g, dtype=torch.float64) * (slen - length - 2).to(torch.float64)).to(torch.int64)
    ar = torch.arange(length, device=dev)[None, :]
    has_indel = torch.rand(n, device=dev, generator=g) < indel_read_frac
    is_del = torch.rand(n, device=dev, generator=g) < 0.5
    p = torch.randint(10, length - 10, (n,), device=dev, generator=g)[:, None]
    # deletion in read: offsets skip one reference base from p on; insertion: a random base at p
    adj = torch.where(has_indel[:, None] & is_del[:, None] & (ar >= p), 1, 0) - \
        torch.where(has_indel[:, None] & (~is_del[:, None]) & (ar > p), 1, 0)
    src = (sop_t[seq] + pos)[:, None] + ar + adj
    rd = ref[src.reshape(-1)].view(n, length)
    ins = has_indel[:, None] & (~is_del[:, None]) & (ar == p)
    rd = torch.where(ins, torch.randint(0, 4, (n, length), dtype=torch.uint8, device=dev, generator=g), rd)
    mut = torch.rand((n, length), device=dev, generator=g) < sub_rate
    rd = torch.where(mut, (rd + torch.randint(1, 4, (n, length), dtype=torch.uint8, device=dev, generator=g)) & 3, rd)
    strand = torch.rand(n, device=dev, generator=g) < 0.5
    rc = (3 - rd).flip(1)
    rd = torch.where(strand[:, None], rc, rd)
    table = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    ascii_reads = table[rd.to(torch.int64)].contiguous().view(-1)
    truth = torch.stack([seq, pos, strand.to(torch.int64)], dim=1)
    return ascii_reads, truth


def _noisy_reads(ref, start, rc, length, g, sub_rate, indel_read_frac):
    """reads of `length` bases read off the concatenated reference at global offsets `start` (forward orientation), then
    reverse-complemented where rc: substitutions, and one 1-base indel in a fraction of the reads (as make_reads_gpu)"""
    dev = ref.device
    n = start.numel()
    ar = torch.arange(length, device=dev)[None, :]
    has_indel = torch.rand(n, device=dev, generator=g) < indel_read_frac
    is_del = torch.rand(n, device=dev, generator=g) < 0.5
    p = torch.randint(10, length - 10, (n,), device=dev, generator=g)[:, None]
    adj = torch.where(has_indel[:, None] & is_del[:, None] & (ar >= p), 1, 0) - \
        torch.where(has_indel[:, None] & (~is_del[:, None]) & (ar > p), 1, 0)
    rd = ref[(start[:, None] + ar + adj).reshape(-1)].view(n, length)
    ins = has_indel[:, None] & (~is_del[:, None]) & (ar == p)
    rd = torch.where(ins, torch.randint(0, 4, (n, length), dtype=torch.uint8, device=dev, generator=g), rd)
    mut = torch.rand((n, length), device=dev, generator=g) < sub_rate
    rd = torch.where(mut, (rd + torch.randint(1, 4, (n, length), dtype=torch.uint8, device=dev, generator=g)) & 3, rd)
    rd = torch.where(rc[:, None], (3 - rd).flip(1), rd)
    table = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    return table[rd.to(torch.int64)].contiguous().view(-1)


def make_pairs_gpu(ref: torch.Tensor, sop: np.ndarray, n: int, length: int, seed: int, insert_mean: float = 300.0, insert_sd: float = 30.0,
                   sub_rate: float = 0.01, indel_read_frac: float = 0.02):
    """Paired-end reads, FR orientation (BASELINE configs[2]): fragments of length N(insert_mean,
    insert_sd) at uniform loci, either strand.  -> (ASCII reads [n*length], ASCII mates [n*length], truth int64 [n,4] =
    (seq, fragment start, strand, fragment length)); all on the device."""
    dev = ref.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    nseq = len(sop) - 1
    seq = torch.randint(0, nseq, (n,), device=dev, generator=g)
    sop_t = torch.as_tensor(np.asarray(sop, dtype=np.int64), device=dev)
    flen = torch.clamp((torch.randn(n, device=dev, generator=g) * insert_sd + insert_mean).round().to(torch.int64), min=length)
    slen = sop_t[seq + 1] - sop_t[seq]
    pos = (torch.rand(n, device=dev, generator=g, dtype=torch.float64) * (slen - flen - 4).to(torch.float64)).to(torch.int64)
    strand = torch.rand(n, device=dev, generator=g) < 0.5
    left = sop_t[seq] + pos                                   # first `length` bases of the fragment's forward strand
    right = sop_t[seq] + pos + flen - length                  # its last `length` bases
    # forward fragment: read = left, mate = revcomp(right); reverse fragment: read = revcomp(right), mate = left
    r1 = _noisy_reads(ref, torch.where(strand, right, left), strand, length, g, sub_rate, indel_read_frac)
    r2 = _noisy_reads(ref, torch.where(strand, left, right), ~strand, length, g, sub_rate, indel_read_frac)
    return r1, r2, torch.stack([seq, pos, strand.to(torch.int64), flen], dim=1)
