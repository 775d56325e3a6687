#!/usr/bin/env python3
"""bench.py -- mapped reads/s of the seed-and-extend hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path (k-mer seeding -> candidate binning -> Smith-Waterman
score pass -> banded traceback) over one batch of synthetic reads that is already resident in
HBM.  Default workload = BASELINE.json configs[1]: 1 M x 150 bp single-end Illumina-shape reads
vs a 3 Gbp synthetic reference (24 x 125 Mbp, 15 % repeat content), k=13 s=6.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); rank 0 builds the index and
broadcasts the image over xGMI; every rank then maps its own shard of reads (weak scaling, no
data-path collective).

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
OPS_PER_CELL = 8.6                                # VALU ops per Gotoh cell of the packed kernel: 8.6 instructions per cell pair
                                                  # (perm, add, max3, add, 2 x (add, max3), 0.6 for the running maximum), each counting as 2 cell-ops


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nchr", type=int, default=24)
    ap.add_argument("--chr-mbp", type=float, default=125.0)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--sub-batch", type=int, default=262144)
    ap.add_argument("--streams", type=int, default=1, help="mappers (HIP streams) that take the sub-batches in turn: kernels of consecutive sub-batches overlap")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-buffers", action="store_true", help="also time smaltgpu_map_batch on pageable host buffers (PCIe-inclusive rate, extra field)")
    ap.add_argument("--cpu-sample", type=int, default=150000)
    return ap.parse_args()


def cpu_baseline(args, ref_pack, idx, pos, sop, names, k, s, reads_ascii, nreads, rlen, gix=None):
    """Time the UNMODIFIED reference (`oracle/_ref/smalt map -n T`) on a bounded sample of the same
    reads against the same index (kind "reference"); falls back to the oracle port."""
    smalt = os.path.join(ROOT, "oracle", "_ref", "smalt")
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get('SMALT_BENCH_CPU_THREADS', '16'))))   # the GPU box's CPU share per GPU
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
            return dict(value=rate, unit="mapped reads/s", cores=cores, kind="reference",
                        sample="smalt map -n %d on the first %d vs %d reads of the bench batch (slope: index load cancels), same index files; %.1f s + %.1f s wall"
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


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from smalt_amd import api, gpuindex, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl = RCCL over xGMI; SMALT_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 path
        dist.init_process_group(backend=os.environ.get("SMALT_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    k, s = 13, 6
    chrlen = int(args.chr_mbp * 1e6)
    nchr = args.nchr
    sop = np.arange(nchr + 1, dtype=np.int64) * chrlen
    names = ["chr%d" % (i + 1) for i in range(nchr)]
    tot = int(sop[-1])

    # ---- setup (untimed): reference + index image in HBM; rank 0 builds, RCCL broadcast ----
    t0 = time.time()
    image = {}
    # small references (4^k > 2 * bases / s) get the collision-type index: every rank builds it from the broadcast
    # reference with the library's own builder (smaltgpu_index_build_device); the default 3 Gbp reference gets the perfect
    # type and its image is broadcast as a whole
    native_build = 4 ** k > 2 * (tot // s)
    if rank == 0:
        ref = gpuindex.make_reference_gpu(nchr, chrlen, 20261004, dev)
        if native_build:
            image = {"ref": ref}
        else:
            packed = gpuindex.pack_reference(ref)
            idx, pos = gpuindex.build_perfect_index(ref, sop, k, s)
            image = {"idx": idx, "pos": pos, "packed": packed, "ref": ref}
    image, bcast_s = shard.broadcast_image(image, dev, 0, order=("ref",) if native_build else ("idx", "pos", "packed", "ref"))     # RCCL over xGMI for N > 1
    ref = image["ref"]
    idx, pos, packed = (None, None, None) if native_build else (image["idx"], image["pos"], image["packed"])
    bcast_ms = bcast_s * 1e3
    gix = None
    if native_build:
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        ascii_ref = lut[ref.long()]
        gix = api.Index.build_device(ascii_ref.data_ptr(), [int(x) for x in sop], names, k, s, local)
        del ascii_ref
    # reads of this rank (weak scaling: every rank maps args.reads reads of its own)
    torch.cuda.synchronize()
    t_idx = time.time() - t0
    reads_ascii, _ = gpuindex.make_reads_gpu(ref, sop, args.reads, args.read_len, 777 + rank)
    del ref
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    if rank == 0:
        print("[bench] setup: reference+index %.1f s, reads %.1f s" % (t_idx, setup_s - t_idx), file=sys.stderr, flush=True)

    if gix is None:
        desc = api.IndexDesc()
        desc.k, desc.s, desc.typ, desc.nbits_key, desc.nbits_lo = k, s, 0, 2 * k, 0
        desc.npos, desc.nwords = int(pos.numel()), 0
        desc.idx, desc.pos, desc.packed = idx.data_ptr(), pos.data_ptr(), packed.data_ptr()
        desc.wordidx = desc.posidx = None
        desc.nseq = nchr
        sop_u64 = np.ascontiguousarray(sop.astype(np.uint64))
        desc.sop = sop_u64.ctypes.data
        desc.on_device = 1
        gix = api.Index.from_desc(desc, local)
    par = gix.default_params()
    sub = min(args.sub_batch, args.reads)
    os.environ.setdefault("SMALTGPU_CANDS_PER_READ", "768")
    mappers = [api.Mapper(gix, sub, args.read_len) for _ in range(max(1, args.streams))]
    mapper = mappers[0]
    offs = torch.arange(sub + 1, dtype=torch.int64, device=dev) * args.read_len
    torch.cuda.synchronize()

    ms_acc, work_acc = {}, [0] * 32

    def one_step(collect):
        mapped = total_res = 0

        def fetch(mp, n):
            nonlocal mapped, total_res
            out = mp.fetch_end()
            st = np.ctypeslib.as_array(C.cast(out.stat, C.POINTER(C.c_uint32)), shape=(n, 8))
            mapped += int((st[:, 7] > 0).sum())
            total_res += int(out.res_off[n])
            if collect:
                ms, wk = mp.timers()
                for kk, v in ms.items():
                    ms_acc[kk] = ms_acc.get(kk, 0.0) + v
                for i in range(32):
                    work_acc[i] += wk[i]
        state = {}                        # mapper -> reads of its batch in flight (kernels enqueued, results not yet fetched)
        for bi, b0 in enumerate(range(0, args.reads, sub)):
            n = min(sub, args.reads - b0)
            mp = mappers[bi % len(mappers)]
            prev = state.pop(mp, None)
            if prev is not None:
                mp.fetch_begin()                      # waits for the mapper's previous sub-batch, enqueues the copies of its results
            mp.map_batch_device(reads_ascii.data_ptr() + b0 * args.read_len, 0, offs.data_ptr(), n, n * args.read_len, par)
            state[mp] = n
            if prev is not None:
                fetch(mp, prev)                       # results put in read order on the host while the device runs the new sub-batch
        for mp, n in list(state.items()):
            mp.fetch_begin()
            fetch(mp, n)
        return mapped, total_res

    import ctypes as C

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tw = time.time()
        one_step(False)
        if rank == 0:
            print("[bench] warmup step %.2f s" % (time.time() - tw), file=sys.stderr, flush=True)
    barrier()
    t1 = time.time()
    mapped = 0
    for _ in range(args.steps):
        m_, _r = one_step(True)
        mapped += m_
    barrier()
    dt = time.time() - t1
    dt, (mapped_all,) = shard.reduce_step(dt, [mapped], dev)      # max over ranks, whole-job count

    if rank == 0:
        value = mapped_all / dt
        nlaunch = args.steps * ((args.reads + sub - 1) // sub)
        dom = max(ms_acc, key=lambda kk: ms_acc[kk])
        cells = work_acc[2]
        sw_ms = ms_acc.get("sw_full", 0.0)
        sw_tops = cells * OPS_PER_CELL / (sw_ms * 1e-3) / 1e12 if sw_ms > 0 else 0.0
        seed_bytes = work_acc[0] * 8 + work_acc[1] * 12
        seed_ms = ms_acc.get("seed", 0.0) + ms_acc.get("cands", 0.0)
        seed_gbs = seed_bytes / (seed_ms * 1e-3) / 1e9 if seed_ms > 0 else 0.0
        traffic_sw = traffic_seed = None          # HBM bytes per launch from the committed PMC passes (profiles/), scaled to this sub-batch
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            sc = sub / float(pt["reads_per_launch"])
            traffic_sw = (pt["kernels"]["sw_full"]["fetch"] + pt["kernels"]["sw_full"]["write"]) * sc
            traffic_seed = sum(pt["kernels"][kk]["fetch"] + pt["kernels"][kk]["write"] for kk in ("seed", "cands")) * sc
        except Exception:
            pass
        roof_sw = dict(kernel="k_sw_full16", bound="valu", achieved=sw_tops, peak=VALU_PEAK_TOPS, unit="TOP/s",
                       frac=sw_tops / VALU_PEAK_TOPS, traffic=traffic_sw, gcups=cells / (sw_ms * 1e-3) / 1e9 if sw_ms > 0 else 0.0,
                       avg_launch_ms=sw_ms / nlaunch, cells_per_launch=cells / nlaunch)
        roof_seed = dict(kernel="k_seed+k_cands", bound="hbm", achieved=seed_gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                         frac=seed_gbs / HBM_PEAK_GBS, traffic=traffic_seed, avg_launch_ms=seed_ms / nlaunch,
                         bytes_per_launch=seed_bytes / nlaunch)
        line = {
            "metric": "mapped reads/sec (1Mx150bp vs 3Gbp ref)", "value": value, "unit": "mapped reads/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",      # K2a: packed half floats holding exact integer scores
            "config": {"workload": "configs[1]: %d x %d bp single-end reads per GPU vs %d x %.0f Mbp synthetic reference (15%% repeats), k=%d s=%d, best-only"
                       % (args.reads, args.read_len, nchr, args.chr_mbp, k, s),
                       "reads_per_gpu_per_step": args.reads, "sub_batch": sub, "mapped_fraction": mapped_all / (world * args.steps * args.reads),
                       "reads_per_s_total": world * args.steps * args.reads / dt, "setup_s": setup_s, "index_broadcast_ms": bcast_ms,
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
            "cands_per_read": work_acc[5] / max(1, world * args.steps * args.reads), "kept_per_read": work_acc[6] / max(1, world * args.steps * args.reads),
            "long_window_tasks_per_read": work_acc[17] / max(1, world * args.steps * args.reads), "ranked_per_read": work_acc[3] / max(1, world * args.steps * args.reads), "scored_in_reference_order_per_read": work_acc[4] / max(1, world * args.steps * args.reads),
            "hits_per_read": work_acc[1] / max(1, world * args.steps * args.reads),
        }
        if args.host_buffers:       # the boundary's host-buffer entry point: H2D of the reads + D2H of the results inside the timing
            hb = reads_ascii[:sub * args.read_len].cpu().numpy()
            ho = (np.arange(sub + 1, dtype=np.uint64) * np.uint64(args.read_len))
            mapper.map_batch_raw(hb, ho, None, par)
            th = time.time()
            mapper.map_batch_raw(hb, ho, None, par)
            line["host_buffers"] = {"reads_per_s": sub / (time.time() - th), "reads": sub, "note": "PCIe-inclusive, not `value`"}
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
            try:
                line["cpu_baseline"] = cpu_baseline(args, None if native_build else packed.cpu().numpy(), None if native_build else idx.cpu().numpy(),
                                                    None if native_build else pos.cpu().numpy(), sop, names, k, s,
                                                    reads_ascii.cpu().numpy(), args.reads, args.read_len, gix)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                line["cpu_baseline"] = dict(value=None, unit="mapped reads/s", cores=0, kind="reference", sample="failed: %r" % (e,))
        print(json.dumps(line))
    for mp_ in mappers:
        mp_.close()
    gix.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
