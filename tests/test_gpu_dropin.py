"""The drop-in claim, end to end: `oracle/_ref/smalt_gpu` is the reference's own `smalt` program with ONE function
replaced -- rmapSingle (src/rmap.c:1648) bound to libsmaltgpu through the C ABI (integration/rmap_gpu.c); FASTQ input,
result post-processing (sorting, pruning, mapping qualities), filters and output formatting are the reference's
unmodified code.  `smalt_gpu map` must print byte-for-byte what `smalt map` (the unmodified reference, same build
recipe) prints.  Both binaries are built in the development container from the reference sources where they lie
(oracle/Makefile: ref, ref_gpu); only the binaries travel."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALT = os.path.join(ROOT, "oracle", "_ref", "smalt")
SMALT_GPU = os.path.join(ROOT, "oracle", "_ref", "smalt_gpu")


def _data(tmp, nchr, chrlen, nreads, rlen, seed, with_n=False):
    from smalt_amd import synth
    ch = synth.make_reference(nchr, chrlen, seed=seed, repeat_frac=0.1, n_fam=3, cons_len=300)
    fa, fq = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fq")
    synth.write_fasta(fa, ch)
    reads, _ = synth.make_reads(ch, nreads, rlen, seed=seed + 1, sub_rate=0.02, indel_read_frac=0.2)
    rng = np.random.default_rng(seed + 2)
    with open(fq, "wb") as f:
        for i, r in enumerate(reads):
            b = bytearray(synth.codes_to_ascii(r))
            if with_n and i % 7 == 0:
                b[int(rng.integers(0, len(b)))] = ord("N")
            if i % 11 == 0:
                b = b[:int(rng.integers(10, len(b)))]            # ragged lengths, some shorter than k
            q = bytes(33 + int(x) for x in rng.integers(5, 41, size=len(b)))
            f.write(b"@r%d\n" % i + bytes(b) + b"\n+\n" + q + b"\n")
    return fa, fq


@pytest.mark.skipif(not (os.path.exists(SMALT) and os.path.exists(SMALT_GPU)), reason="reference binaries not built (make -C oracle ref ref_gpu)")
@pytest.mark.parametrize("k,s,nchr,chrlen,rlen,opts", [
    (13, 6, 1, 1_000_000, 100, ["-f", "cigar"]),                  # BASELINE configs[0] shape
    (13, 6, 3, 300_000, 150, ["-f", "sam", "-q", "10"]),          # multi-sequence, SAM with mapping qualities, -q
    (11, 3, 2, 200_000, 120, ["-f", "cigar", "-d", "-1"]),        # all alignments (no best-only)
    (13, 6, 2, 250_000, 150, ["-f", "cigar", "-x", "-c", "0.4"]),  # exhaustive search, fractional cover threshold
    (13, 6, 3, 300_000, 150, ["-f", "sam", "-n", "3", "-O", "-r", "-1"]),   # worker threads: blocks of 96 reads per GPU batch; -r -1: no random draws
    (13, 6, 2, 250_000, 100, ["-f", "cigar", "-n", "8", "-O", "-r", "-1", "-x", "-c", "30"]),   # blocks of 256 reads, exhaustive, absolute cover threshold
    (13, 6, 700, 1_500, 100, ["-f", "sam", "-n", "2", "-O", "-r", "-1"]),   # >= 512 reference sequences: concatenated mode (hashCollectHitsUsingCutoff, assignSequenceIndex)
    (11, 4, 600, 2_000, 120, ["-f", "cigar", "-d", "-1"]),                   # concatenated mode, all alignments
    (13, 6, 3000, 400, 100, ["-f", "sam", "-n", "2", "-O", "-r", "-1"]),    # a contig set: 3000 reference sequences (no limit on their number in concatenated mode)
    (13, 6, 3, 300_000, 120, ["-f", "ssaha", "-n", "2", "-O", "-r", "-1"]),  # SSAHA2 lines (fprintREPALIssaha, report.c:579)
    (13, 6, 3, 300_000, 150, ["-f", "sam", "-S", "match=2,subst=-3,gapopen=-5,gapext=-3"]),   # the user's alignment scores (smalt.c:539-550)
])
def test_smalt_map_prints_the_same(k, s, nchr, chrlen, rlen, opts, tmp_path):
    tmp = str(tmp_path)
    if "-r" not in opts:
        opts = opts + ["-r", "3"]          # the default -r 0 seeds the tie-breaking draws with the time of day (randef.h:19)
    fa, fq = _data(tmp, nchr, chrlen, 1200, rlen, seed=k * 100 + s + nchr, with_n=True)
    pre = os.path.join(tmp, "idx")
    subprocess.run([SMALT, "index", "-k", str(k), "-s", str(s), pre, fa], check=True, capture_output=True)
    out_ref, out_gpu = os.path.join(tmp, "ref.out"), os.path.join(tmp, "gpu.out")
    subprocess.run([SMALT, "map"] + opts + ["-o", out_ref, pre, fq], check=True, capture_output=True)
    env = dict(os.environ, SMALTGPU_INDEX_PREFIX=pre)
    a = [ln for ln in open(out_ref).read().split("\n") if not ln.startswith("@PG")]
    assert sum(1 for ln in a if ln and not ln.startswith("@")) >= 1000
    # batched binding (the worker maps a block of reads per GPU batch) and per-read binding (rmapSingle, one read per call)
    # ... with pools of 8 ranked candidates per read (overflow: the library re-maps in smaller batches) and with two index
    # images / mapper sets fed from one queue (SMALTGPU_NDEV=2; on a one-GPU box both on the same device, the second image a
    # device-to-device copy of the first)
    for extra in ({}, {"SMALTGPU_PER_READ": "1"}, {"SMALTGPU_CANDS_PER_READ": "8"}, {"SMALTGPU_NDEV": "2", "SMALTGPU_COMBINE_READS": "256"}):
        r = subprocess.run([SMALT_GPU, "map"] + opts + ["-o", out_gpu, pre, fq], capture_output=True, env=dict(env, **extra))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        b = [ln for ln in open(out_gpu).read().split("\n") if not ln.startswith("@PG")]
        assert len(a) == len(b)
        diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
        assert not diff, (extra, diff[:3])
    # the same command line through smaltgpu-map, the program made of the library alone (ingest, GPU path, post-processing,
    # report: SURVEY 8f N4 + N1); -O (keep the input order) is what it always does
    prog = os.path.join(ROOT, "smalt_amd", "smaltgpu-map")
    r = subprocess.run([prog] + [o for o in opts if o != "-O"] + ["-B", "500", "-o", out_gpu, pre, fq], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    b = [ln for ln in open(out_gpu).read().split("\n") if not ln.startswith("@PG")]
    assert len(a) == len(b)
    diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
    assert not diff, ("smaltgpu-map", diff[:3])


@pytest.mark.skipif(not (os.path.exists(SMALT) and os.path.exists(SMALT_GPU)), reason="reference binaries not built (make -C oracle ref ref_gpu)")
def test_binding_goes_through_the_library_and_threads(tmp_path):
    """Without the index prefix the bound binary cannot map (so the equality above is not the CPU path in disguise), and
    with worker threads (-n 3: one RMap = one mapper = one HIP stream per thread, -O keeps input order) it still prints
    the reference's lines for reads with a unique best alignment (ties are drawn with a thread-shared drand48)."""
    tmp = str(tmp_path)
    fa, fq = _data(tmp, 2, 300_000, 900, 100, seed=77)
    pre = os.path.join(tmp, "idx")
    subprocess.run([SMALT, "index", "-k", "13", "-s", "6", pre, fa], check=True, capture_output=True)
    env = dict(os.environ)
    env.pop("SMALTGPU_INDEX_PREFIX", None)
    r = subprocess.run([SMALT_GPU, "map", "-f", "cigar", "-o", os.path.join(tmp, "x.out"), pre, fq], capture_output=True, env=env)
    assert r.returncode != 0
    out_ref, out_gpu = os.path.join(tmp, "ref.out"), os.path.join(tmp, "gpu.out")
    subprocess.run([SMALT, "map", "-f", "cigar", "-o", out_ref, pre, fq], check=True, capture_output=True)
    env["SMALTGPU_INDEX_PREFIX"] = pre
    r = subprocess.run([SMALT_GPU, "map", "-n", "3", "-O", "-f", "cigar", "-o", out_gpu, pre, fq], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    a, b = open(out_ref).read().split("\n"), open(out_gpu).read().split("\n")
    assert len(a) == len(b)
    same = sum(1 for x, y in zip(a, b) if x == y)
    assert same >= 0.97 * len(a), (same, len(a))
    for x, y in zip(a, b):
        if x != y:                       # only the reads whose best alignment is not unique may differ (class R / mapq 0-3)
            assert x.split()[0].split(":")[1] in ("R", "S") or int(x.split()[0].split(":")[2]) <= 3, (x, y)


@pytest.mark.skipif(not (os.path.exists(SMALT) and os.path.exists(SMALT_GPU)), reason="reference binaries not built (make -C oracle ref ref_gpu)")
def test_a_read_the_reference_fails_on_fails_here_too(tmp_path):
    """Error behaviour: with a gap extension much cheaper than the opening (-S ...,gapopen=-5,gapext=-2) the score
    pass (textbook recurrence, swsimd.c) and the banded traceback pass (restricted recurrence, alignment.c:1029) can disagree
    on a candidate; the reference stops with ERRCODE_SWATSCOR at that read (rmap.c:1417).  Both programs on the library must
    stop at the same read instead of printing something -- the bound program with the reference's own message."""
    tmp = str(tmp_path)
    fa, fq = _data(tmp, 3, 300_000, 1200, 150, seed=1309, with_n=True)
    pre = os.path.join(tmp, "idx")
    subprocess.run([SMALT, "index", "-k", "13", "-s", "6", pre, fa], check=True, capture_output=True)
    opts = ["-f", "sam", "-S", "match=2,subst=-3,gapopen=-5,gapext=-2", "-r", "3"]
    r = subprocess.run([SMALT, "map"] + opts + ["-o", os.path.join(tmp, "ref.out"), pre, fq], capture_output=True, text=True)
    assert r.returncode != 0 and "Inconsistency when calculating Smith-Waterman scores" in r.stdout + r.stderr
    where = [ln for ln in (r.stdout + r.stderr).split("\n") if "when processing read No." in ln][0].strip()
    env = dict(os.environ, SMALTGPU_INDEX_PREFIX=pre)
    g = subprocess.run([SMALT_GPU, "map"] + opts + ["-o", os.path.join(tmp, "gpu.out"), pre, fq], capture_output=True, text=True, env=env)
    assert g.returncode != 0, "the bound program went on where the reference stops"
    assert "Inconsistency when calculating Smith-Waterman scores" in g.stdout + g.stderr and where in g.stdout + g.stderr, (where, (g.stdout + g.stderr)[-1500:])
    n = subprocess.run([os.path.join(ROOT, "smalt_amd", "smaltgpu-map")] + opts + ["-o", os.path.join(tmp, "nat.out"), pre, fq], capture_output=True, text=True)
    name = where.split("'")[1]
    assert n.returncode != 0 and name in n.stderr, (name, n.stderr[-1500:])


def _pair_data(tmp, nchr, chrlen, npairs, rlen, seed, rep=0.3, ins=(300, 30)):
    from smalt_amd import synth
    ch = synth.make_reference(nchr, chrlen, seed=seed, repeat_frac=rep, n_fam=2, cons_len=400, divergence=0.03)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ch)
    r1, r2, _ = synth.make_pairs(ch, npairs, rlen, seed=seed + 1, insert_mean=ins[0], insert_sd=ins[1], sub_rate=0.02, indel_read_frac=0.2)
    rng = np.random.default_rng(seed + 2)
    fqs = []
    for which, reads in ((1, r1), (2, r2)):
        fq = os.path.join(tmp, "reads_%d.fq" % which)
        with open(fq, "wb") as f:
            for i, r in enumerate(reads):
                b = bytearray(synth.codes_to_ascii(r))
                u = rng.random()
                if u < 0.03:
                    b = bytearray(synth.codes_to_ascii(rng.integers(0, 4, size=len(b), dtype=np.uint8)))    # maps nowhere
                elif u < 0.05:
                    b = b[:int(rng.integers(5, 16))]                                                        # around the word length
                elif u < 0.10:
                    b = b[:int(rng.integers(24, len(b)))]
                if rng.random() < 0.05 and len(b) > 4:
                    b[int(rng.integers(0, len(b)))] = ord("N")
                q = bytes(33 + int(x) for x in rng.integers(5, 41, size=len(b)))
                f.write(b"@p%d/%d\n" % (i, which) + bytes(b) + b"\n+\n" + q + b"\n")
        fqs.append(fq)
    return fa, fqs


@pytest.mark.skipif(not (os.path.exists(SMALT) and os.path.exists(SMALT_GPU)), reason="reference binaries not built (make -C oracle ref ref_gpu)")
@pytest.mark.parametrize("k,s,nchr,chrlen,rlen,ins,opts", [
    (13, 6, 3, 300_000, 100, (300, 30), ["-f", "cigar", "-i", "500"]),                          # BASELINE configs[2] shape in small
    (13, 6, 3, 300_000, 150, (300, 30), ["-f", "sam", "-i", "500", "-q", "10"]),                # SAM with mapping qualities and flags, -q
    (11, 3, 2, 200_000, 75, (350, 40), ["-f", "sam", "-i", "600", "-j", "100", "-l", "mp"]),    # mate-pair library, insert range
    (13, 6, 3, 300_000, 100, (300, 30), ["-f", "cigar", "-i", "500", "-n", "3", "-O", "-r", "-1"]),   # worker threads
    (13, 6, 600, 2_000, 100, (300, 30), ["-f", "sam", "-i", "500"]),                            # >= 512 reference sequences: concatenated mode
    (13, 6, 2, 250_000, 100, (300, 30), ["-f", "cigar", "-i", "500", "-x"]),                    # exhaustive search (long hit info for both mates)
    (13, 6, 2, 250_000, 100, (300, 30), ["-f", "cigar", "-i", "500", "-x", "-c", "0.4"]),       # cover threshold (fraction of each mate) in all four rounds
    (13, 6, 3, 300_000, 150, (300, 30), ["-f", "sam", "-i", "500", "-x", "-c", "45"]),          # cover threshold in bases (the reference accepts -c with -x only)
    (13, 6, 600, 2_000, 100, (300, 30), ["-f", "cigar", "-i", "500", "-x", "-c", "0.5"]),       # cover threshold in concatenated mode: plain rounds take the sequential candidate stage
    (13, 6, 3, 300_000, 100, (300, 30), ["-f", "ssaha", "-i", "500"]),                          # SSAHA2 lines with the pair classes
    (13, 6, 3, 300_000, 100, (300, 30), ["-f", "cigar", "-i", "500", "-S", "subst=-3,gapopen=-6"]),   # the user's alignment scores in all rounds
])
def test_smalt_map_pairs_prints_the_same(k, s, nchr, chrlen, rlen, ins, opts, tmp_path):
    """Paired reads: rmapPair's rounds (rare mate, restricted mate, unrestricted re-map, re-map over the on-the-fly k=5 index)
    run on the GPU for whole blocks of pairs (integration/rmap_gpu.c: rmapGpuPairBatch); pairing, mapping qualities, pair
    classes and output are the reference's own code.  Byte-identical output to the unmodified `smalt map`."""
    tmp = str(tmp_path)
    if "-r" not in opts:
        opts = opts + ["-r", "3"]          # the default -r 0 seeds the tie-breaking draws with the time of day (randef.h:19)
    fa, fqs = _pair_data(tmp, nchr, chrlen, 700, rlen, seed=k * 131 + s + nchr, rep=0.3 if nchr < 100 else 0.0, ins=ins)
    pre = os.path.join(tmp, "idx")
    subprocess.run([SMALT, "index", "-k", str(k), "-s", str(s), pre, fa], check=True, capture_output=True)
    out_ref, out_gpu = os.path.join(tmp, "ref.out"), os.path.join(tmp, "gpu.out")
    subprocess.run([SMALT, "map"] + opts + ["-o", out_ref, pre] + fqs, check=True, capture_output=True)
    env = dict(os.environ, SMALTGPU_INDEX_PREFIX=pre)
    a = [ln for ln in open(out_ref).read().split("\n") if not ln.startswith("@PG")]
    assert sum(1 for ln in a if ln and not ln.startswith("@")) >= 1000
    for extra in ({}, {"SMALTGPU_CANDS_PER_READ": "8"}):
        r = subprocess.run([SMALT_GPU, "map"] + opts + ["-o", out_gpu, pre] + fqs, capture_output=True, env=dict(env, **extra))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        b = [ln for ln in open(out_gpu).read().split("\n") if not ln.startswith("@PG")]
        assert len(a) == len(b)
        diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
        assert not diff, (extra, len(diff), diff[:3])
    # the same command line through smaltgpu-map, the program made of the library alone: the rounds, the decisions between them,
    # pairing and the paired report are the library's own; blocks of 250 pairs
    prog = os.path.join(ROOT, "smalt_amd", "smaltgpu-map")
    r = subprocess.run([prog] + [o for o in opts if o != "-O"] + ["-B", "250", "-o", out_gpu, pre] + fqs, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    b = [ln for ln in open(out_gpu).read().split("\n") if not ln.startswith("@PG")]
    assert len(a) == len(b)
    diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
    assert not diff, ("smaltgpu-map", len(diff), diff[:3])
    # negative control: with the mates of another pair the output must differ (the comparison above is not vacuous)
    rot = os.path.join(tmp, "rot_2.fq")
    recs = open(fqs[1]).read().split("\n")
    with open(rot, "w") as f:
        f.write("\n".join(recs[4:4 * 700] + recs[:4]) + "\n")
    r = subprocess.run([prog] + [o for o in opts if o != "-O"] + ["-o", out_gpu, pre, fqs[0], rot], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    b = [ln for ln in open(out_gpu).read().split("\n") if not ln.startswith("@PG")]
    assert sum(1 for x, y in zip(a, b) if x != y) > 100


@pytest.mark.skipif(not (os.path.exists(SMALT) and os.path.exists(SMALT_GPU)), reason="reference binaries not built (make -C oracle ref ref_gpu)")
def test_pairs_with_32_workers_and_blocks_of_4096(tmp_path):
    """More waiting worker threads than mapper slots x cohort size: `smalt_gpu map -n 32` with blocks of 4096 pairs
    (SMALTGPU_BLOCK_READS) over 140 000 pairs -- 32 workers queue their rounds (hit totals, A, B, C, D) in the shared batch
    queue (integration/gpu_combine.c) at once.  The output must be the unmodified program's, line for line."""
    tmp = str(tmp_path)
    fa, fqs = _pair_data(tmp, 3, 300_000, 140_000, 100, seed=4321, rep=0.3)
    pre = os.path.join(tmp, "idx")
    subprocess.run([SMALT, "index", "-k", "13", "-s", "6", pre, fa], check=True, capture_output=True)
    out_ref, out_gpu = os.path.join(tmp, "ref.out"), os.path.join(tmp, "gpu.out")
    opts = ["-f", "cigar", "-i", "500", "-O", "-r", "-1"]
    subprocess.run([SMALT, "map", "-n", "16"] + opts + ["-o", out_ref, pre] + fqs, check=True, capture_output=True)
    env = dict(os.environ, SMALTGPU_INDEX_PREFIX=pre, SMALTGPU_BLOCK_READS="4096")
    r = subprocess.run([SMALT_GPU, "map", "-n", "32"] + opts + ["-o", out_gpu, pre] + fqs, capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    a, b = open(out_ref).read().split("\n"), open(out_gpu).read().split("\n")
    assert len(a) == len(b) and len(a) > 250_000
    diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
    assert not diff, (len(diff), diff[:3])


def satellite_pairs(tmp, copies=620):
    """Reference with a satellite-like repeat (`copies` units) and read pairs on it: (index prefix, [fastq 1, fastq 2], pairs)."""
    rng = np.random.default_rng(99)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rnd = lambda n: acgt[rng.integers(0, 4, size=n)].tobytes()
    ua, ub = rnd(150), rnd(150)
    unit = ua + rnd(80) + ub                                   # read 1 lies in ua, read 2 in ub, 230 bases downstream
    chr1 = b"".join(rnd(int(rng.integers(650, 800))) + unit for _ in range(copies)) + rnd(500)
    chr2 = b"".join(rnd(int(rng.integers(300, 400))) + ub for _ in range(copies + 80)) + rnd(500)      # ub is the more frequent one: ua's mate is mapped first
    chr3 = rnd(200_000)
    fa = os.path.join(tmp, "ref.fa")
    with open(fa, "wb") as f:
        for i, c in enumerate((chr1, chr2, chr3)):
            f.write(b">chr%d\n" % (i + 1))
            for o in range(0, len(c), 70):
                f.write(c[o:o + 70] + b"\n")
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    pairs = []
    for i in range(12):                                        # repeat pairs, some with a substitution
        a, b = bytearray(ua[20:120]), bytearray(ub[25:125][::-1].translate(comp))
        if i % 3 == 1:
            a[40] = ord("A") if a[40] != ord("A") else ord("C")
        pairs.append((bytes(a), bytes(b)))
    for i in range(60):                                        # ordinary pairs from the unique sequence
        p = int(rng.integers(0, len(chr3) - 400))
        pairs.append((chr3[p:p + 100], chr3[p + 200:p + 300][::-1].translate(comp)))
    fqs = [os.path.join(tmp, "r_%d.fq" % w) for w in (1, 2)]
    for w in (0, 1):
        with open(fqs[w], "wb") as f:
            for i, pr in enumerate(pairs):
                f.write(b"@p%d/%d\n" % (i, w + 1) + pr[w] + b"\n+\n" + b"I" * len(pr[w]) + b"\n")
    pre = os.path.join(tmp, "idx")
    subprocess.run([SMALT, "index", "-k", "13", "-s", "6", pre, fa], check=True, capture_output=True)
    return pre, fqs, pairs


@pytest.mark.skipif(not (os.path.exists(SMALT) and os.path.exists(SMALT_GPU)), reason="reference binaries not built (make -C oracle ref ref_gpu)")
def test_pairs_with_more_than_1023_search_intervals(tmp_path):
    """A first mate with 1300 equally good alignments (a satellite-like repeat, exhaustive search `-x`: the depth cut is the
    maximum depth 2048) gives its mate 1300 search intervals in the restricted round -- one per alignment (rmap.c:354-436; the
    reference has no limit, interval.c:98-121) -- and the k = 5 round over them gathers > 100 k hits per strand.  The interval
    number has 11 bits in the hit sort key of a restricted call: the pairs map, and both programs print what the unmodified
    `smalt map` prints.  The binding reports the largest interval count it saw (SMALTGPU_TIMING), which keeps this test honest."""
    tmp = str(tmp_path)
    pre, fqs, pairs = satellite_pairs(tmp, copies=1300)
    out_ref, out_gpu = os.path.join(tmp, "ref.out"), os.path.join(tmp, "gpu.out")
    opts = ["-f", "cigar", "-i", "500", "-r", "-1", "-x"]
    subprocess.run([SMALT, "map"] + opts + ["-o", out_ref, pre] + fqs, check=True, capture_output=True)
    a = open(out_ref).read().split("\n")
    assert len(a) == 2 * len(pairs) + 1
    prog = os.path.join(ROOT, "smalt_amd", "smaltgpu-map")
    env = dict(os.environ, SMALTGPU_INDEX_PREFIX=pre)
    # third run: the binding's limit lowered below this input's 1266 intervals -- such pairs are not an error of the block, the
    # reference's own rmapPair maps them on the worker's thread (integration/rmap_gpu.c: on_cpu)
    for cmd, e in (([prog] + opts + ["-o", out_gpu, pre] + fqs, os.environ), ([SMALT_GPU, "map"] + opts + ["-o", out_gpu, pre] + fqs, dict(env, SMALTGPU_TIMING="1")),
                   ([SMALT_GPU, "map"] + opts + ["-o", out_gpu, pre] + fqs, dict(env, SMALTGPU_MAX_INTERVALS="1000", SMALTGPU_TIMING="1"))):
        r = subprocess.run(cmd, capture_output=True, env=e)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        if "SMALTGPU_TIMING" in e:
            left = [ln for ln in r.stderr.decode().split("\n") if "pairs left to the reference" in ln]
            most = [ln for ln in r.stderr.decode().split("\n") if "most search intervals" in ln]
            assert most and int(most[0].rsplit(":", 1)[1]) > 1023, most
            nleft = int(left[0].rsplit(":", 1)[1])
            assert (nleft >= 1) if "SMALTGPU_MAX_INTERVALS" in e else (nleft == 0), (left, most)       # with the lowered limit the repeat pairs take the CPU route
        b = open(out_gpu).read().split("\n")
        diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
        assert len(a) == len(b) and not diff, (cmd[0], len(diff), diff[:3])


@pytest.mark.skipif(not os.path.exists(SMALT), reason="reference binary not built (make -C oracle ref)")
def test_serial_order_mode_prints_what_a_serial_run_prints(tmp_path):
    """`smalt map -n 0` keeps one hit list for all reads, and the list only grows with the longest read so far (hashhit.c:1280):
    the fixture of tests/golden/make_golden_history.py has 100-base reads whose alignments depend on whether the 260-base read
    came before them.  SMALTGPU_SERIAL_ORDER=1 makes smaltgpu-map carry that length from read to read (in blocks of 16 reads
    here, so that it is carried across blocks and workers); the output is that of the unmodified program."""
    import gzip
    import json
    import golden_util as gu
    entry = json.load(open(os.path.join(gu.GOLD, "manifest_history.json")))
    tmp = str(tmp_path)
    paths = {}
    for ext in ("fa", "fq"):
        paths[ext] = os.path.join(tmp, "h." + ext)
        with gzip.open(os.path.join(gu.GOLD, "%s.%s.gz" % (entry["tag"], ext)), "rb") as g, open(paths[ext], "wb") as f:
            f.write(g.read())
    pre = os.path.join(tmp, "idx")
    subprocess.run([SMALT, "index", "-k", str(entry["k"]), "-s", str(entry["s"]), pre, paths["fa"]], check=True, capture_output=True)
    opts = ["-f", "cigar", "-r", "-1", "-d", "-1"] + entry["opts"].split()
    out_ref, out_gpu = os.path.join(tmp, "ref.out"), os.path.join(tmp, "gpu.out")
    subprocess.run([SMALT, "map"] + opts + ["-o", out_ref, pre, paths["fq"]], check=True, capture_output=True)
    a = open(out_ref).read().split("\n")
    prog = os.path.join(ROOT, "smalt_amd", "smaltgpu-map")
    outs = {}
    for mode in ("1", "0"):
        env = dict(os.environ, SMALTGPU_SERIAL_ORDER=mode)
        r = subprocess.run([prog] + opts + ["-B", "16", "-o", out_gpu, pre, paths["fq"]], capture_output=True, env=env)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs[mode] = open(out_gpu).read().split("\n")
    diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, outs["1"])) if x != y]
    assert len(a) == len(outs["1"]) and not diff, (len(diff), diff[:3])
    assert outs["0"] != a                       # the fixture does depend on the order
