#!/usr/bin/env python3
"""bench.py -- mapped reads/s of the seed-and-extend hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path (k-mer seeding -> candidate binning -> Smith-Waterman
score pass -> banded traceback) over one batch of synthetic reads that is already resident in
HBM.  Default workload = BASELINE.json configs[1]: 1 M x 150 bp single-end Illumina-shape reads
vs a 3 Gbp synthetic reference (24 x 125 Mbp, 15 % repeat content), k=13 s=6.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  `python bench.py --gpus N` starts
the N ranks itself (before anything touches HIP); under `torch.distributed.run` it takes the ranks it
is given.  Rank 0 builds the index with the library's own builder and broadcasts the image over xGMI;
the ranks then pull sub-batches of the job's reads (BASELINE.json configs[3]: 20 M reads on 8 GPUs =
2.5 M per GPU) from one shared cursor -- the analogue of the reference's workers pulling read blocks
from one queue (threads.c:548) -- with no data-path collective (weak scaling: the job grows with N).

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12     # lane-ops/s: 256 CU x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md); a packed 16-bit
                                                  # instruction issues every 4 cycles for 64 lanes x 2 halves = the same 32 cell-ops/clk/SIMD
HBM_PEAK_GBS = 8000.0
OPS_PER_CELL_SURVEY = 12.0                        # scalar integer ops per Gotoh cell (1 add, 5 max, 3 sub/add, 1 compare, 2 select/move)
OPS_PER_CELL = 8.6                                # VALU ops per Gotoh cell of the packed kernel: 8.6 instructions per cell pair
                                                  # (perm, add, max3, add, 2 x (add, max3), 0.6 for the running maximum), each counting as 2 cell-ops


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nchr", type=int, default=24)
    ap.add_argument("--chr-mbp", type=float, default=125.0)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU and step (default: 1 M at N = 1 = configs[1]; 2.5 M at N > 1 = configs[3]'s 20 M / 8)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--sub-batch", type=int, default=262144)
    ap.add_argument("--streams", type=int, default=1, help="mappers (HIP streams) that take the sub-batches in turn: kernels of consecutive sub-batches overlap")
    ap.add_argument("--paired", action="store_true", help="BASELINE configs[2]: 1 M read PAIRS (2 x read-len, FR, fragments N(300,30), -i 500) through smaltgpu_map_pairs_resident: "
                    "rmapPair's rounds on the GPU, the decisions between them in the library; pairs/s with kernel ms per round, roofline and cpu_baseline")
    ap.add_argument("--long", action="store_true", help="BASELINE configs[4] shape: PacBio-shape reads of 8 kbp (3 %% substitutions, 5 %% insertions, 4 %% deletions) vs the 3 Gbp "
                    "reference, k=20 s=13; --reads of them (default 3000 of the config's 100 k: a step must finish within minutes)")
    ap.add_argument("--static-shards", action="store_true", help="N > 1: contiguous shard per rank instead of the shared sub-batch cursor")
    ap.add_argument("--long-batch", type=int, default=3000, help="--long: reads per device batch (K3 of 8 kbp reads is bound by the latency of single reads: 1000 per batch 418 reads/s, 3000 per batch 444)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-buffers", action="store_true", help="skip the extra PCIe-inclusive measurement of smaltgpu_map_batch on pageable host buffers")
    ap.add_argument("--cpu-sample", type=int, default=150000)
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) from a parent that has not
    touched HIP (no torch import yet) and leave with their status.  Rank 0 prints the JSON line."""
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            st = p.poll()
            if st is None:
                continue
            live.remove(p)
            if st != 0 and rc == 0:
                rc = st
                for q in live:            # a rank died: the others would wait in a collective for ever
                    q.terminate()
    return rc


def host_core_count():
    """every core this process may run on: CPU affinity and the container's CPU quota (cgroup v2 cpu.max)"""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return cores


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, ref_pack, idx, pos, sop, names, k, s, reads_ascii, nreads, rlen, gix=None):
    """Time the UNMODIFIED reference (`oracle/_ref/smalt map -n T`) on a bounded sample of the same
    reads against the same index (kind "reference"); falls back to the oracle port."""
    smalt = os.path.join(ROOT, "oracle", "_ref", "smalt")
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:                                                     # a container's CPU quota counts too (cgroup v2: "<quota> <period>" or "max")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    host_cores = cores                                       # every core this process may run on (the box's share per GPU)
    cores = max(1, int(os.environ.get('SMALT_BENCH_CPU_THREADS', cores)))
    from smalt_amd import indexfile
    n2 = min(args.cpu_sample, nreads)
    n1 = max(n2 // 10, 1000)
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        prefix = os.path.join(tmp, "bench")
        tot = int(sop[-1])
        if idx is None:
            gix.save(prefix)                       # index built by the library: its own writer
        else:
            indexfile.write_sma(prefix, names, sop, ref_pack)
            indexfile.write_smi_perfect(prefix, k, s, idx, pos, (tot + s - 1) // s - 1)
        rd = reads_ascii[: n2 * rlen].reshape(n2, rlen)

        def write_fq(path, n):
            with open(path, "wb") as f:
                q = b"I" * rlen
                for i in range(n):
                    f.write(b"@r%d\n" % i + rd[i].tobytes() + b"\n+\n" + q + b"\n")
        f1, f2 = os.path.join(tmp, "s1.fq"), os.path.join(tmp, "s2.fq")
        write_fq(f1, n1)
        write_fq(f2, n2)
        if os.path.exists(smalt):
            def run(fq):
                t = time.time()
                subprocess.run([smalt, "map", "-n", str(cores), "-f", "cigar", "-o", os.path.join(tmp, "o.cig"), prefix, fq],
                               check=True, capture_output=True)
                return time.time() - t
            t1 = run(f1)
            t2 = run(f2)
            # cigar output has one line per mapped read and none for unmapped ones
            mapped = len({ln.split()[1] for ln in open(os.path.join(tmp, "o.cig")) if ln.startswith("cigar:")})
            dt = max(t2 - t1, 1e-6)      # the index load (same in both runs) cancels
            rate = (n2 - n1) / dt * (mapped / n2)
            return dict(value=rate, unit="mapped reads/s", cores=cores, host_cores=host_cores, cpu_model=cpu_model(), kind="reference",
                        sample="smalt map -n %d (all cores this process may use) on the first %d vs %d reads of the bench batch (slope: index load cancels), same index files; %.1f s + %.1f s wall"
                        % (cores, n1, n2, t1, t2))
        # port: the oracle's C restatement, single thread
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        oix = ol.lib().or_index_read(prefix.encode())
        m = ol.Mapper(oix)
        par = ol.default_params(oix)
        n = min(2000, n2)
        t = time.time()
        mapped = 0
        for i in range(n):
            _, res = m.map(rd[i].tobytes(), None, par)
            mapped += bool(res)
        dt = time.time() - t
        m.close()
        return dict(value=mapped / dt, unit="mapped reads/s", cores=1, kind="port",
                    sample="oracle C restatement, 1 thread, first %d reads of the bench batch, %.1f s" % (n, dt))


class _DevArray:
    """A device array of the library seen through __cuda_array_interface__, so that torch.distributed can broadcast the
    index image the library built without a copy."""

    def __init__(self, ptr, n, typestr="<i4"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def cpu_baseline_pairs(args, gix, reads, mates, rlen, cores_info):
    """The UNMODIFIED reference (`oracle/_ref/smalt map -n T -i 500`) on a bounded sample of the same pairs and the same index
    files; the index load cancels in the slope of two sample sizes."""
    smalt = os.path.join(ROOT, "oracle", "_ref", "smalt")
    if not os.path.exists(smalt):
        return dict(value=None, unit="read pairs/s", cores=0, kind="reference", sample="oracle/_ref/smalt is not built")
    cores, host_cores = cores_info
    n2 = min(max(4000, args.cpu_sample // 3), reads.shape[0])
    n1 = max(n2 // 10, 500)
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        prefix = os.path.join(tmp, "bench")
        gix.save(prefix)

        def write_fq(path, arr, n, which):
            q = b"I" * rlen
            with open(path, "wb") as f:
                for i in range(n):
                    f.write(b"@p%d/%d\n" % (i, which) + arr[i].tobytes() + b"\n+\n" + q + b"\n")

        def run(n):
            f1, f2 = os.path.join(tmp, "a_%d.fq" % n), os.path.join(tmp, "b_%d.fq" % n)
            write_fq(f1, reads, n, 1)
            write_fq(f2, mates, n, 2)
            t = time.time()
            subprocess.run([smalt, "map", "-n", str(cores), "-i", "500", "-f", "cigar", "-o", os.path.join(tmp, "o.cig"), prefix, f1, f2], check=True, capture_output=True)
            return time.time() - t
        t1, t2 = run(n1), run(n2)
        both = 0                                          # pairs with both mates mapped, as `value` counts them
        prev = None
        for ln in open(os.path.join(tmp, "o.cig")):
            f = ln.split()
            mapped = f[0].split(":")[1] not in ("N", "R")
            if prev is not None and prev[0] == f[1].rsplit("/", 1)[0]:
                both += prev[1] and mapped
                prev = None
            else:
                prev = (f[1].rsplit("/", 1)[0], mapped)
        rate = (n2 - n1) / max(t2 - t1, 1e-6) * (both / n2)
        return dict(value=rate, unit="read pairs/s", cores=cores, host_cores=host_cores, cpu_model=cpu_model(), kind="reference",
                    sample="smalt map -n %d -i 500 on the first %d vs %d pairs of the bench batch (slope: index load cancels), same index files; %.1f s + %.1f s wall"
                    % (cores, n1, n2, t1, t2))


def paired_main(args):
    """BASELINE configs[2]: read pairs through the library alone.  One step = smaltgpu_map_pairs_resident over every block of
    the job: the rounds of rmapPair (rmap.c:1744) on the GPU, the decisions between the rounds (post-call passes, proper-pair
    probe, search intervals) on host threads.  Reads and mates are resident in HBM when the timed region starts; the pairing
    and the report lines (smaltgpu_report_emit_pairs) belong to the whole-program figure, not to this path figure."""
    import ctypes as C

    import torch
    from smalt_amd import api, gpuindex
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    k, s, rlen = 13, 6, args.read_len
    chrlen, nchr = int(args.chr_mbp * 1e6), args.nchr
    sop = np.arange(nchr + 1, dtype=np.int64) * chrlen
    names = ["chr%d" % (i + 1) for i in range(nchr)]
    npairs = args.reads or 1_000_000
    t0 = time.time()
    ref = gpuindex.make_reference_gpu(nchr, chrlen, 20261004, dev)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    ascii_ref = lut[ref.long()]
    gix = api.Index.build_device(ascii_ref.data_ptr(), [int(x) for x in sop], names, k, s, 0)
    del ascii_ref
    reads, mates, _ = gpuindex.make_pairs_gpu(ref, sop, npairs, rlen, 777)
    del ref
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    print("[bench] setup: reference, index (%.1f ms on the device) and %d pairs in %.1f s" % (gix.build_ms, npairs, setup_s), file=sys.stderr, flush=True)
    host_cores = host_core_count()
    cores = max(1, int(os.environ.get("SMALT_BENCH_CPU_THREADS", host_cores)))
    par = gix.default_params()
    sub = min(args.sub_batch, npairs)
    os.environ.setdefault("SMALTGPU_CANDS_PER_READ", "768")
    import threading
    # the timed region runs on ONE mapper (HIP stream) so that the event duration of a kernel is that of a kernel that has the
    # device to itself (the roofline figure); a second mapper serves the extra two-stream measurement behind it
    nstream = max(1, args.streams)
    nmappers = max(nstream, 2)
    mappers = [api.Mapper(gix, sub, rlen, slot_budget_gb=64 // nmappers) for _ in range(nmappers)]
    offs = torch.arange(sub + 1, dtype=torch.int64, device=dev) * rlen
    h_offs = np.arange(sub + 1, dtype=np.uint64) * np.uint64(rlen)
    po = api.PairOpts(0, 500, api.LIB_PE, 0, max(1, cores // nstream))
    L = api.lib()
    handles = [C.c_void_p(L.smaltgpu_pairs_create()) for _ in range(nmappers)]
    L.smaltgpu_pairs_host_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    host_ms = np.zeros(8)
    nblk = (npairs + sub - 1) // sub
    kms = np.zeros((5, 16))
    work = np.zeros((5, 32), dtype=np.uint64)
    calls = np.zeros(4, dtype=np.uint64)
    round_ms = np.zeros(4)
    lock = threading.Lock()

    def one_step(collect, nstream=nstream):
        """the blocks of the job, taken in turn by `nstream` host threads with a mapper (HIP stream) each: the host work between
        the rounds of one block runs while the device works on the other block"""
        tot = {"both": 0, "any": 0, "next": 0, "err": None}

        def drive(si):
            mp, handle = mappers[si], handles[si]
            while True:
                with lock:
                    b = tot["next"]
                    tot["next"] += 1
                if b >= nblk or tot["err"]:
                    return
                b0 = b * sub
                n = min(sub, npairs - b0)
                src = api.ResidentReads()
                for w, t in enumerate((reads, mates)):
                    src.d_bases[w] = t.data_ptr() + b0 * rlen
                    src.d_quals[w] = None
                    src.d_read_off[w] = offs.data_ptr()
                    src.read_off[w] = h_offs.ctypes.data
                    src.nreads[w] = n
                rv = L.smaltgpu_map_pairs_resident(mp.h, C.byref(src), None, None, None, None, n, C.byref(par), C.byref(po), handle)
                if rv:
                    tot["err"] = "smaltgpu_map_pairs_resident failed: %s" % L.smaltgpu_last_error().decode()
                    return
                npr = C.c_uint32()
                info = C.POINTER(api.PairInfo)()
                cl = (C.c_uint64 * 4)()
                rm = (C.c_double * 4)()
                L.smaltgpu_pairs_info(handle, C.byref(npr), C.byref(info), cl, rm)
                a = np.ctypeslib.as_array(C.cast(info, C.POINTER(C.c_uint16)), shape=(n, 3))       # flags+rounds | nali[0] | nali[1]
                km = (C.c_double * 80)()
                wk = (C.c_uint64 * 160)()
                L.smaltgpu_pairs_timers(handle, km, wk)
                with lock:
                    tot["both"] += int(((a[:, 1] > 0) & (a[:, 2] > 0)).sum())
                    tot["any"] += int(((a[:, 1] > 0) | (a[:, 2] > 0)).sum())
                    if collect:
                        kms[:] += np.array(km).reshape(5, 16)
                        work[:] += np.array(wk, dtype=np.uint64).reshape(5, 32)
                        calls[:] += np.array(cl, dtype=np.uint64)
                        round_ms[:] += np.array(rm)
                        hm = (C.c_double * 8)()
                        L.smaltgpu_pairs_host_times(handle, hm, 8)
                        host_ms[:] += np.array(hm)
        th = [threading.Thread(target=drive, args=(i,)) for i in range(nstream)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if tot["err"]:
            raise SystemExit(tot["err"])
        return tot["both"], tot["any"]

    for w in range(args.warmup):
        tw = time.time()
        one_step(False)
        print("[bench] warmup step %.2f s" % (time.time() - tw), file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    t1 = time.time()
    both = any_ = 0
    for _ in range(args.steps):
        b_, a_ = one_step(True)
        both += b_
        any_ += a_
    torch.cuda.synchronize()
    dt = time.time() - t1
    ntim = L.smaltgpu_timers(mappers[0].h, None, None, 0)
    tn = [L.smaltgpu_timer_name(i).decode() for i in range(ntim)]
    sw = tn.index("sw_full")
    cells = float(work[:, 2].sum())
    sw_ms = float(kms[:, sw].sum())
    tcups = cells / (sw_ms * 1e-3) / 1e12 if sw_ms > 0 else 0.0
    nlaunch = int((calls > 0).sum()) * nblk * args.steps or 1          # K2a launches: one per round and block
    rname = ["A first mate", "B second mate restricted", "C second mate again", "D first mate over the on-the-fly index", "hit totals"]
    per_round = {rname[r]: {tn[i]: float(kms[r, i]) / args.steps for i in range(ntim) if kms[r, i] > 0} for r in range(5)}
    gpu_ms = float(kms.sum()) / args.steps
    line = {
        "metric": "mapped read pairs/sec (1M pairs 2x%dbp vs 3Gbp ref)" % rlen, "value": both / dt, "unit": "read pairs/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": "configs[2]: %d read pairs 2 x %d bp (FR, fragments N(300,30), -i 500) vs %d x %.0f Mbp synthetic reference (15%% repeats), k=%d s=%d; "
                               "smaltgpu_map_pairs_resident: rmapPair's rounds on the GPU, the decisions between them on %d host threads" % (npairs, rlen, nchr, args.chr_mbp, k, s, cores),
                   "pairs_per_step": npairs, "block_pairs": sub, "pairs_per_s_total": args.steps * npairs / dt,
                   "both_mates_mapped_fraction": both / (args.steps * npairs), "a_mate_mapped_fraction": any_ / (args.steps * npairs),
                   "mapping_calls_per_pair": {rname[r]: float(calls[r]) / (args.steps * npairs) for r in range(4)}, "setup_s": setup_s, "index_build_ms": gix.build_ms,
                   "host_threads": cores, "streams": nstream},
        "roofline": dict(kernel="k_sw_full16", bound="valu", achieved=tcups * OPS_PER_CELL, peak=VALU_PEAK_TOPS, unit="TOP/s", frac=tcups * OPS_PER_CELL / VALU_PEAK_TOPS,
                         frac_issue=tcups * OPS_PER_CELL / VALU_PEAK_TOPS, ops_per_cell_issue=OPS_PER_CELL, frac_survey=tcups * OPS_PER_CELL_SURVEY / VALU_PEAK_TOPS,
                         ops_per_cell_survey=OPS_PER_CELL_SURVEY, gcups=tcups * 1e3, traffic=None, avg_launch_ms=sw_ms / nlaunch, cells_per_launch=cells / nlaunch,
                         peak_note="256 CU x 4 SIMD x 32 lane-ops/clk x 2.4 GHz; same units as the single-end line; all rounds' K2a launches together; with 2 streams the kernels of two blocks share the device, so the per-launch time includes that sharing"),
        "kernel_ms_per_step_by_round": per_round,
        "cands_phase_share_by_round": {rname[r]: [round(float(x) / max(1.0, float(work[r, 8:17].sum())), 4) for x in work[r, 8:17]] for r in range(5) if work[r, 8:17].sum() > 0},
        "cands_phase_ticks_per_read_by_round": {rname[r]: [round(float(x) / max(1.0, float(calls[r])), 1) for x in list(work[r, 8:17]) + [work[r, 21]]] for r in range(4) if calls[r] > 0},
        "hits_per_read_by_round": {rname[r]: float(work[r, 1]) / max(1.0, float(calls[r]) if r < 4 else 1.0) for r in range(4) if calls[r] > 0},
        "kernel_ms_per_step": {tn[i]: float(kms[:, i].sum()) / args.steps for i in range(ntim)},
        "gpu_busy_fraction": gpu_ms / (dt * 1e3 / args.steps),
        "round_wall_ms_per_step": {rname[r]: float(round_ms[r]) / args.steps for r in range(4)},
        "host_ms_per_step": dict(zip(["behind A: post-call pass + search intervals", "behind B: post-call pass", "proper-pair probe", "behind C: post-call pass", "plan of D",
                                      "behind D: post-call pass", "hit totals (wall, incl. the device)", "whole calls (wall)"], [float(x) / args.steps for x in host_ms])),
    }
    # the same job on two mappers (HIP streams) taking the blocks in turn: the host work between the rounds of one block runs
    # while the device works on the other block.  Not `value`: kernels of two blocks share the device, so their event durations
    # are no longer those of a kernel on its own.
    if nstream == 1:
        one_step(False, 2)
        torch.cuda.synchronize()
        t2 = time.time()
        b2, _ = one_step(False, 2)
        torch.cuda.synchronize()
        line["two_streams"] = {"pairs_per_s": b2 / (time.time() - t2), "note": "blocks dealt to two mappers; one extra step, not `value`"}
    if not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline_pairs(args, gix, reads.cpu().numpy().reshape(npairs, rlen), mates.cpu().numpy().reshape(npairs, rlen), rlen, (cores, host_cores))
        except Exception as e:
            line["cpu_baseline"] = dict(value=None, unit="read pairs/s", cores=0, kind="reference", sample="failed: %r" % (e,))
    print(json.dumps(line), flush=True)
    for h in handles:
        L.smaltgpu_pairs_free(h)
    for mp in mappers:
        mp.close()
    gix.close()


def long_main(args):
    """BASELINE configs[4] shape.  One step = the whole path over `--reads` long reads resident in HBM, in batches of `--long-batch` reads
    (the scratch of a long read is large: direction matrices, hit slots).  The dominant kernel is the un-banded score pass in its
    strip form (k_sw_strip16: reads and windows beyond the register tiling); its roofline is the same VALU issue peak as the
    headline kernel's, cells = read length x window length per ranked candidate."""
    import ctypes as C

    import torch
    from smalt_amd import api, gpuindex, synth
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    k, s, rlen = 20, 13, 8000
    chrlen, nchr = int(args.chr_mbp * 1e6), args.nchr
    sop = [i * chrlen for i in range(nchr + 1)]
    names = ["chr%d" % (i + 1) for i in range(nchr)]
    nreads = args.reads or 3000
    t0 = time.time()
    ref = gpuindex.make_reference_gpu(nchr, chrlen, 20261004, dev)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    ascii_ref = torch.cat([lut[c.long()] for c in ref.split(1 << 28)])
    gix = api.Index.build_device(ascii_ref.data_ptr(), sop, names, k, s, 0)
    del ascii_ref
    rng = np.random.default_rng(4242)
    reads = []
    for i in range(nreads):               # sources fetched from HBM, sequencing errors applied on the host (smalt_amd/synth.py)
        c = int(rng.integers(0, nchr))
        p = int(rng.integers(0, chrlen - rlen - 8))
        src = ref[c * chrlen + p: c * chrlen + p + rlen + 4].cpu().numpy()
        r, _ = synth.make_long_reads([src], 1, rlen, seed=1000 + i, sub=0.03, ins=0.05, dele=0.04)
        reads.append(synth.codes_to_ascii(r[0]))
    del ref
    torch.cuda.empty_cache()
    maxlen = max(len(r) for r in reads)
    batch = min(max(1, args.long_batch), nreads)
    lens = np.array([len(r) for r in reads], dtype=np.int64)
    off = np.zeros(nreads + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    d_bases = torch.from_numpy(np.frombuffer(b"".join(reads), dtype=np.uint8).copy()).to(dev)
    par = gix.default_params()
    mapper = api.Mapper(gix, batch, maxlen)
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    print("[bench] setup: reference, index (%.1f ms on the device) and %d reads of ~%d bases in %.1f s" % (gix.build_ms, nreads, rlen, setup_s), file=sys.stderr, flush=True)
    ms_acc, work_acc = {}, [0] * 32

    def one_step(collect):
        mapped = 0
        for b0 in range(0, nreads, batch):
            n = min(batch, nreads - b0)
            d_off = torch.from_numpy(off[b0:b0 + n + 1] - off[b0]).to(dev)
            mapper.map_batch_device(d_bases.data_ptr() + int(off[b0]), 0, d_off.data_ptr(), n, int(off[b0 + n] - off[b0]), par)
            try:
                out = mapper.fetch_results()
            except api.SmaltGpuError as e:
                if e.code not in (-5, -6, -8):
                    raise
                out = None                 # single reads over a device-side limit keep their error code; the others are complete
            if out is not None:
                st = np.ctypeslib.as_array(C.cast(out.stat, C.POINTER(C.c_uint32)), shape=(n, 10))
                mapped += int((st[:, 7] > 0).sum())
            if collect:
                ms, wk = mapper.timers()
                for kk, v in ms.items():
                    ms_acc[kk] = ms_acc.get(kk, 0.0) + v
                for i in range(32):
                    work_acc[i] += wk[i]
        return mapped
    for w in range(args.warmup):
        tw = time.time()
        one_step(False)
        print("[bench] warmup step %.2f s" % (time.time() - tw), file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    t1 = time.time()
    mapped = 0
    for _ in range(args.steps):
        mapped += one_step(True)
    torch.cuda.synchronize()
    dt = time.time() - t1
    nlaunch = args.steps * ((nreads + batch - 1) // batch)
    cells = float(work_acc[2])
    sw_ms = ms_acc.get("sw_scalar", 0.0) + ms_acc.get("sw_full", 0.0)       # strips and bands run in the sw_scalar bracket
    tcups = cells / (sw_ms * 1e-3) / 1e12 if sw_ms > 0 else 0.0
    line = {
        "metric": "mapped reads/sec (8 kbp PacBio-shape reads vs 3Gbp ref, k=20)", "value": mapped / dt, "unit": "mapped reads/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "configs[4] shape: %d of the config's 100 k reads, %d-base sources with 3 %% substitutions, 5 %% insertions, 4 %% deletions vs %d x %.0f Mbp "
                               "synthetic reference (15 %% repeats), k=%d s=%d (collision-type index built by the library), batches of %d reads" % (nreads, rlen, nchr, args.chr_mbp, k, s, batch),
                   "reads_per_step": nreads, "mean_read_len": float(lens.mean()), "mapped_fraction": mapped / (args.steps * nreads), "bases_per_s": float(lens.sum()) * args.steps / dt,
                   "setup_s": setup_s, "index_build_ms": gix.build_ms, "ranked_per_read": work_acc[3] / (args.steps * nreads)},
        "roofline": dict(kernel="k_sw_strip16", bound="valu", achieved=tcups * OPS_PER_CELL, peak=VALU_PEAK_TOPS, unit="TOP/s", frac=tcups * OPS_PER_CELL / VALU_PEAK_TOPS,
                         ops_per_cell_issue=OPS_PER_CELL, frac_survey=tcups * OPS_PER_CELL_SURVEY / VALU_PEAK_TOPS, gcups=tcups * 1e3, traffic=None, avg_launch_ms=sw_ms / nlaunch,
                         cells_per_launch=cells / nlaunch, peak_note="same VALU issue peak and ops per cell as the headline kernel; the strip form carries H/F between strips through LDS and HBM"),
        "kernel_ms_per_step": {kk: v / args.steps for kk, v in ms_acc.items()},
        "align_phase": {"ticks_window_band_trace": work_acc[24:27], "band_passes": work_acc[27], "aligned_candidates": work_acc[28], "band_width_sum": work_acc[31]},
    }
    if not args.no_cpu_baseline:
        smalt = os.path.join(ROOT, "oracle", "_ref", "smalt")
        cores = max(1, int(os.environ.get("SMALT_BENCH_CPU_THREADS", host_core_count())))
        try:
            with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
                prefix = os.path.join(tmp, "ix")
                gix.save(prefix)

                def run(n):
                    pth = os.path.join(tmp, "r%d.fq" % n)
                    with open(pth, "wb") as f:
                        for i in range(n):
                            f.write(b"@r%d\n" % i + reads[i] + b"\n+\n" + b"5" * len(reads[i]) + b"\n")
                    tt = time.time()
                    subprocess.run([smalt, "map", "-n", str(cores), "-f", "cigar", "-o", os.path.join(tmp, "o.cig"), prefix, pth], check=True, capture_output=True)
                    return time.time() - tt
                n1, n2 = 4, min(nreads, max(20, cores + 8))
                ta, tb = run(n1), run(n2)
                line["cpu_baseline"] = dict(value=(n2 - n1) / max(tb - ta, 1e-6), unit="mapped reads/s", cores=cores, host_cores=host_core_count(), cpu_model=cpu_model(), kind="reference",
                                            sample="smalt map -n %d on the first %d vs %d reads (slope: index load cancels), same index files; %.1f s + %.1f s wall" % (cores, n1, n2, ta, tb))
        except Exception as e:
            line["cpu_baseline"] = dict(value=None, unit="mapped reads/s", cores=0, kind="reference", sample="failed: %r" % (e,))
    print(json.dumps(line), flush=True)
    mapper.close()
    gix.close()


def main():
    args = parse_args()
    if args.long:
        if args.gpus > 1:
            raise SystemExit("--long measures one GPU (configs[4])")
        long_main(args)
        return
    if args.paired:
        if args.gpus > 1:
            raise SystemExit("--paired measures one GPU (configs[2]); the multi-GPU line is the single-end job (configs[3])")
        paired_main(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))            # before torch / HIP are touched in this process
    import ctypes as C

    import torch
    import torch.distributed as dist
    from smalt_amd import api, gpuindex, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()           # does not initialise HIP
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl = RCCL over xGMI, one rank per GPU.  With fewer GPUs than ranks (rehearsal on a one-GPU box) the ranks share
        # devices and RCCL cannot be used: gloo carries the same collectives.  SMALT_BENCH_BACKEND overrides.
        backend = os.environ.get("SMALT_BENCH_BACKEND", "nccl" if ndev >= world else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if os.environ.get("SMALT_BENCH_DRYRUN"):   # tests/test_shard_gloo.py: launcher, rendezvous and the shared cursor without a device
        dealer = shard.BatchDealer(23, static=args.static_shards)
        if world > 1:
            dist.barrier()                              # as ahead of the timed region of the real run: no rank deals before all are there
        dealer.start("dry")
        got = []
        while True:
            j = dealer.next()
            if j is None:
                break
            got.append(j)
            time.sleep(0.003)                           # a sub-batch takes time: the other ranks get their turn
        parts = shard.gather_in_rank_order(got)
        # ... and the guided dealing of the N > 1 job: every read of a 1 000 003-read job exactly once, pieces of 4096 .. 65536
        dealer2 = shard.BatchDealer(1, static=args.static_shards)
        dealer2.start("dry-guided")
        pieces = []
        while True:
            pc = dealer2.next_range(1000003, 65536, 4096)
            if pc is None:
                break
            pieces.append(pc)
            time.sleep(0.0005 * (1 + rank))            # ranks of different speed
        guided = shard.gather_in_rank_order(pieces)
        if rank == 0:
            print(json.dumps({"n_gpus": world, "dry_run": True, "backend": backend, "dealt": parts, "guided": guided}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    local = local % max(1, ndev)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    k, s = 13, 6
    chrlen = int(args.chr_mbp * 1e6)
    nchr = args.nchr
    sop = np.arange(nchr + 1, dtype=np.int64) * chrlen
    names = ["chr%d" % (i + 1) for i in range(nchr)]
    tot = int(sop[-1])
    per_gpu = args.reads or (1_000_000 if world == 1 else 2_500_000)      # configs[1]; configs[3]: 20 M reads / 8 GPUs
    job_reads = per_gpu * world

    # ---- setup (untimed): reference + index image in HBM.  Rank 0 builds the index with the library's builder
    # (smaltgpu_index_build_device) and broadcasts the image (RCCL over xGMI for N > 1) ----
    t0 = time.time()
    image, meta, gix0, build_ms = {}, None, None, 0.0
    if rank == 0:
        ref = gpuindex.make_reference_gpu(nchr, chrlen, 20261004, dev)
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        ascii_ref = lut[ref.long()]
        gix0 = api.Index.build_device(ascii_ref.data_ptr(), [int(x) for x in sop], names, k, s, local)
        build_ms = gix0.build_ms
        del ascii_ref
        if world > 1:
            d0 = gix0.info()
            nkeys = 1 << (2 * k if d0.typ == 0 else d0.nbits_key)
            image = {"idx": torch.as_tensor(_DevArray(d0.idx, nkeys + 1), device=dev), "pos": torch.as_tensor(_DevArray(d0.pos, d0.npos), device=dev),
                     "packed": torch.as_tensor(_DevArray(d0.packed, tot // 10 + 1), device=dev), "ref": ref}
            if d0.typ != 0:
                image["wordidx"] = torch.as_tensor(_DevArray(d0.wordidx, d0.nwords + 1), device=dev)
                image["posidx"] = torch.as_tensor(_DevArray(d0.posidx, d0.nwords + 1), device=dev)
            meta = [d0.typ, d0.nbits_key, d0.nbits_lo, d0.npos, d0.nwords]
    bcast_ms = 0.0
    bcast_by_array = {}
    if world > 1:
        ml = [meta]
        dist.broadcast_object_list(ml, 0)
        meta = ml[0]
        order = ("idx", "pos", "packed", "ref") + (("wordidx", "posidx") if meta[0] != 0 else ())
        image, bcast_s = shard.broadcast_image(image, dev, 0, order=order)
        bcast_ms = bcast_s * 1e3
        bcast_by_array = {kk: {"ms": v["seconds"] * 1e3, "GB_per_s": v["bytes"] / max(v["seconds"], 1e-9) / 1e9} for kk, v in shard.broadcast_image.last_by_array.items()}
        ref = image["ref"]
    if rank == 0:
        gix = gix0
    else:                                       # the other ranks adopt the broadcast arrays (kept alive by `image`)
        desc = api.IndexDesc()
        desc.k, desc.s, desc.typ, desc.nbits_key, desc.nbits_lo, desc.npos, desc.nwords = k, s, meta[0], meta[1], meta[2], meta[3], meta[4]
        desc.idx, desc.pos, desc.packed = image["idx"].data_ptr(), image["pos"].data_ptr(), image["packed"].data_ptr()
        desc.wordidx = image["wordidx"].data_ptr() if meta[0] != 0 else None
        desc.posidx = image["posidx"].data_ptr() if meta[0] != 0 else None
        desc.nseq = nchr
        sop_u64 = np.ascontiguousarray(sop.astype(np.uint64))
        desc.sop = sop_u64.ctypes.data
        desc.on_device = 1
        gix = api.Index.from_desc(desc, local)
    torch.cuda.synchronize()
    t_idx = time.time() - t0
    # Reads of the whole job on every rank (3 GB for 20 M reads), generated in chunks of one GPU's share: any rank can
    # then take any sub-batch from the shared cursor.  Chunk 0 is the N = 1 workload.
    if args.static_shards:
        chunks = {rank: gpuindex.make_reads_gpu(ref, sop, per_gpu, args.read_len, 777 + rank)[0]}
    else:
        chunks = {c: gpuindex.make_reads_gpu(ref, sop, per_gpu, args.read_len, 777 + c)[0] for c in range(world)}
    del ref
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    if rank == 0:
        print("[bench] setup: reference+index %.1f s (index construction on the device %.1f ms), reads %.1f s" % (t_idx, build_ms, setup_s - t_idx), file=sys.stderr, flush=True)

    par = gix.default_params()
    if os.environ.get("SMALT_BENCH_EXHAUSTIVE"):      # diagnostic, not the BASELINE metric: `smalt map -x` (every seed, deeper candidate lists)
        par.rmapflg |= api.FLG_NOSHRTINFO | api.FLG_SENSITIVE
    sub = min(args.sub_batch, per_gpu)
    os.environ.setdefault("SMALTGPU_CANDS_PER_READ", "768")
    mappers = [api.Mapper(gix, sub, args.read_len) for _ in range(max(1, args.streams))]
    mapper = mappers[0]
    offs = torch.arange(sub + 1, dtype=torch.int64, device=dev) * args.read_len
    torch.cuda.synchronize()
    nsub_chunk = (per_gpu + sub - 1) // sub
    # sub-batch j of the job = sub-batch j % nsub_chunk of chunk j // nsub_chunk; static shards: the rank's own chunk
    dealer = shard.BatchDealer(nsub_chunk * world, static=args.static_shards)

    ms_acc, work_acc = {}, [0] * 32
    mine = {"reads": 0, "launches": 0}

    def one_step(collect, tag):
        mapped = total_res = 0

        def fetch(mp, n):
            nonlocal mapped, total_res
            out = mp.fetch_end()
            st = np.ctypeslib.as_array(C.cast(out.stat, C.POINTER(C.c_uint32)), shape=(n, 10))
            mapped += int((st[:, 7] > 0).sum())
            total_res += int(out.res_off[n])
            if collect:
                ms, wk = mp.timers()
                for kk, v in ms.items():
                    ms_acc[kk] = ms_acc.get(kk, 0.0) + v
                for i in range(32):
                    work_acc[i] += wk[i]
                mine["reads"] += n
                mine["launches"] += 1
        state = {}                        # mapper -> reads of its batch in flight (kernels enqueued, results not yet fetched)
        dealer.start(tag)
        bi = 0
        while True:
            if world > 1:                 # guided dealing over the job's reads: pieces shrink towards the end (shard.BatchDealer.next_range)
                piece = dealer.next_range(job_reads, sub, 32768)
                if piece is None:
                    break
                segs, at, left = [], piece[0], piece[1]
                while left > 0:           # a piece that runs across two chunks of generated reads is mapped as two sub-batches
                    c, b0 = at // per_gpu, at % per_gpu
                    n = min(left, per_gpu - b0)
                    segs.append((c, b0, n))
                    at += n
                    left -= n
            else:
                j = dealer.next()
                if j is None:
                    break
                b0 = (j % nsub_chunk) * sub
                segs = [(j // nsub_chunk, b0, min(sub, per_gpu - b0))]
            for c, b0, n in segs:
                mp = mappers[bi % len(mappers)]
                bi += 1
                prev = state.pop(mp, None)
                if prev is not None:
                    mp.fetch_begin()                      # waits for the mapper's previous sub-batch, enqueues the copies of its results
                mp.map_batch_device(chunks[c].data_ptr() + b0 * args.read_len, 0, offs.data_ptr(), n, n * args.read_len, par)
                state[mp] = n
                if prev is not None:
                    fetch(mp, prev)                       # results put in read order on the host while the device runs the new sub-batch
        for mp, n in list(state.items()):
            mp.fetch_begin()
            fetch(mp, n)
        return mapped, total_res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        tw = time.time()
        one_step(False, "w%d" % w)
        if rank == 0:
            print("[bench] warmup step %.2f s" % (time.time() - tw), file=sys.stderr, flush=True)
    barrier()
    t1 = time.time()
    mapped = 0
    for st_ in range(args.steps):
        m_, _r = one_step(True, "s%d" % st_)
        mapped += m_
    barrier()
    dt = time.time() - t1
    if world == 1:
        mapped_all, reads_min, reads_max = mapped, mine["reads"], mine["reads"]
    else:
        dt, (mapped_all, reads_min, reads_max) = _reduce(shard, dist, torch, dt, mapped, mine["reads"], dev)

    if rank == 0:
        value = mapped_all / dt
        nlaunch = max(1, mine["launches"])            # rank 0's own launches: the kernel figures below are rank 0's
        my_reads = max(1, mine["reads"])
        dom = max(ms_acc, key=lambda kk: ms_acc[kk])
        cells = work_acc[2]
        sw_ms = ms_acc.get("sw_full", 0.0)
        tcups = cells / (sw_ms * 1e-3) / 1e12 if sw_ms > 0 else 0.0
        sw_tops = tcups * OPS_PER_CELL
        seed_bytes = work_acc[0] * 8 + work_acc[1] * 12
        seed_ms = ms_acc.get("seed", 0.0) + ms_acc.get("hits", 0.0) + ms_acc.get("cands", 0.0)
        seed_gbs = seed_bytes / (seed_ms * 1e-3) / 1e9 if seed_ms > 0 else 0.0
        traffic_sw = traffic_seed = None          # HBM bytes per launch from the committed PMC passes (profiles/), scaled to this sub-batch
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            sc = (my_reads / nlaunch) / float(pt["reads_per_launch"])      # mean reads per launch of this run
            traffic_sw = (pt["kernels"]["sw_full"]["fetch"] + pt["kernels"]["sw_full"]["write"]) * sc
            traffic_seed = sum(pt["kernels"][kk]["fetch"] + pt["kernels"][kk]["write"] for kk in ("seed", "cands")) * sc
        except Exception:
            pass
        # Two units for the same time: `frac_issue` counts the VALU instructions the kernel issues per cell (8.6 per cell
        # pair, two cell-ops each; verified in the gfx950 ISA) against the measured issue peak -- the fraction of the machine
        # in use; `frac_survey` counts SURVEY 8(d)'s 12 scalar ops per Gotoh cell against the same peak and exceeds 1 because
        # v_pk_maximum3 does two of those ops (and two cells) per instruction.  `frac` is the issue figure.
        roof_sw = dict(kernel="k_sw_full16", bound="valu", achieved=sw_tops, peak=VALU_PEAK_TOPS, unit="TOP/s",
                       frac=sw_tops / VALU_PEAK_TOPS, frac_issue=sw_tops / VALU_PEAK_TOPS, ops_per_cell_issue=OPS_PER_CELL,
                       frac_survey=tcups * OPS_PER_CELL_SURVEY / VALU_PEAK_TOPS, ops_per_cell_survey=OPS_PER_CELL_SURVEY,
                       peak_note="256 CU x 4 SIMD x 32 lane-ops/clk x 2.4 GHz; packed 16-bit, max3 and perm issue at half rate with two halves each (profiles/r02_valu_rate.txt)",
                       traffic=traffic_sw, gcups=tcups * 1e3, avg_launch_ms=sw_ms / nlaunch, cells_per_launch=cells / nlaunch)
        roof_seed = dict(kernel="k_seed+k_hits+k_cands", bound="hbm", achieved=seed_gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                         frac=seed_gbs / HBM_PEAK_GBS, traffic=traffic_seed, avg_launch_ms=seed_ms / nlaunch,
                         bytes_per_launch=seed_bytes / nlaunch)
        wl = ("configs[1]: %d x %d bp single-end reads" % (per_gpu, args.read_len)) if world == 1 else \
             ("configs[3]: %d x %d bp single-end reads on %d GPUs (%d per GPU)" % (job_reads, args.read_len, world, per_gpu))
        line = {
            "metric": "mapped reads/sec (1Mx150bp vs 3Gbp ref)", "value": value, "unit": "mapped reads/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",      # K2a: packed half floats holding exact integer scores
            "config": {"workload": "%s vs %d x %.0f Mbp synthetic reference (15%% repeats), k=%d s=%d, best-only" % (wl, nchr, args.chr_mbp, k, s),
                       "reads_per_gpu_per_step": per_gpu, "job_reads_per_step": job_reads, "sub_batch": sub,
                       "mapped_fraction": mapped_all / (args.steps * job_reads),
                       "reads_per_s_total": args.steps * job_reads / dt, "setup_s": setup_s, "index_build_ms": build_ms,
                       "index_broadcast_ms": bcast_ms, "index_broadcast_by_array": bcast_by_array, "backend": backend, "rccl_ranks": world if backend == "nccl" else 0,
                       "dealing": "static shards" if args.static_shards or world == 1 else "shared read cursor (c10d store), guided piece sizes 32768..%d" % sub,
                       "reads_taken_min_max_per_rank": [reads_min / args.steps, reads_max / args.steps],
                       "parallelism": "read-shard x%d" % world, "streams": len(mappers)},
            "roofline": roof_sw if dom in ("sw_full",) else roof_seed,
            "roofline_sw": roof_sw, "roofline_seed": roof_seed,
            "kernel_ms_per_step": {kk: v / args.steps for kk, v in ms_acc.items()},
            "dominant_kernel": dom,
            "align_phase": {"ticks_window_band_trace": work_acc[24:27], "band_passes": work_acc[27], "aligned_candidates": work_acc[28], "band_steps": work_acc[29], "sequential_passes": work_acc[30], "band_width_sum": work_acc[31]},
            "cands_phase_share": [round(x / max(1, sum(work_acc[8:17])), 4) for x in work_acc[8:17]],
            "cands_phase_ticks": sum(work_acc[8:17]),
            "cands_windows": {"hits_in_windowed_strands": work_acc[20], "window_gather_share": round(work_acc[21] / max(1, sum(work_acc[8:17])), 4),
                              "table_build_share": round(work_acc[19] / max(1, sum(work_acc[8:17])), 4), "hbm_fallback_strands": work_acc[22]},
            "cands_per_read": work_acc[5] / my_reads, "kept_per_read": work_acc[6] / my_reads,
            "long_window_tasks_per_read": work_acc[17] / my_reads, "ranked_per_read": work_acc[3] / my_reads, "scored_in_reference_order_per_read": work_acc[4] / my_reads,
            "hits_per_read": work_acc[1] / my_reads,
        }
        if not args.no_host_buffers:       # the boundary's host-buffer entry point: H2D of the reads + D2H of the results inside the timing
            hb = chunks[0][:sub * args.read_len].cpu().numpy()
            ho = (np.arange(sub + 1, dtype=np.uint64) * np.uint64(args.read_len))
            mapper.map_batch_raw(hb, ho, None, par)
            th = time.time()
            mapper.map_batch_raw(hb, ho, None, par)
            line["host_buffers"] = {"reads_per_s": sub / (time.time() - th), "reads": sub, "note": "smaltgpu_map_batch on pageable host buffers, PCIe-inclusive, one batch at a time; not `value`"}
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
            try:
                line["cpu_baseline"] = cpu_baseline(args, None, None, None, sop, names, k, s, chunks[0].cpu().numpy(), per_gpu, args.read_len, gix)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                line["cpu_baseline"] = dict(value=None, unit="mapped reads/s", cores=0, kind="reference", sample="failed: %r" % (e,))
        print(json.dumps(line), flush=True)
    for mp_ in mappers:
        mp_.close()
    gix.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _reduce(shard, dist, torch, dt, mapped, my_reads, dev):
    """max step time over ranks, whole-job mapped count, and the spread of reads the ranks took from the cursor"""
    dt, (mapped_all,) = shard.reduce_step(dt, [mapped], dev)
    t = torch.tensor([float(my_reads), -float(my_reads)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return dt, (mapped_all, -float(t[1].item()), float(t[0].item()))


if __name__ == "__main__":
    main()
