#!/usr/bin/env python3
"""End-to-end tool comparison (not the bench.py contract): the reference's `smalt map -n T` against the same program with
its read-mapping worker bound to libsmaltgpu (`oracle/_ref/smalt_gpu`, integration/*.c) on BASELINE configs[1] inputs
written to disk: 3 Gbp reference index (.sma/.smi, written from the GPU-built image), FASTQ of N reads.  Everything
around the hot path -- FASTQ parsing (one reader thread), post-processing and mapping qualities (worker threads), CIGAR
output (one writer) -- is the reference's own host code in both programs.  Wall times are whole-program; the index
load is removed by the difference of two read counts (as bench.py's cpu_baseline does).  Prints one JSON line."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1000000)
    ap.add_argument("--cpu-reads", type=int, default=100000)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--gpu-threads", type=int, default=32, help="worker threads of the bound program (0: same as --threads); about twice the cores: most of them wait for their batch")
    ap.add_argument("--nchr", type=int, default=24)
    ap.add_argument("--chr-mbp", type=float, default=125.0)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--native-repeat", type=int, default=1, help="single-end: run smaltgpu-map this many times on the large file (run-to-run spread)")
    ap.add_argument("--native-gap", type=float, default=0.0, help="seconds to wait before each smaltgpu-map run (the driver clears freed device memory in the background)")
    ap.add_argument("--extra", default="", help="more options for all three programs, e.g. --extra=-p (split reads) or --extra='-x -c 0.3'")
    ap.add_argument("--paired", action="store_true", help="BASELINE configs[2]: read PAIRS (FR, fragments N(300,30), -i 500); --reads counts pairs, rates are pairs/s")
    a = ap.parse_args()
    import torch
    from smalt_amd import gpuindex, indexfile
    dev = torch.device("cuda", 0)
    k, s, rlen = 13, 6, a.read_len
    chrlen = int(a.chr_mbp * 1e6)
    sop = np.arange(a.nchr + 1, dtype=np.int64) * chrlen
    names = ["chr%d" % (i + 1) for i in range(a.nchr)]
    ref = gpuindex.make_reference_gpu(a.nchr, chrlen, 20261004, dev)
    packed = gpuindex.pack_reference(ref)
    idx, pos = gpuindex.build_perfect_index(ref, sop, k, s)
    if a.paired:
        reads, mates, _ = gpuindex.make_pairs_gpu(ref, sop, a.reads, rlen, 777)
        md = mates.cpu().numpy().reshape(a.reads, rlen)
    else:
        reads, _ = gpuindex.make_reads_gpu(ref, sop, a.reads, rlen, 777)
    rd = reads.cpu().numpy().reshape(a.reads, rlen)
    smalt, smalt_gpu = os.path.join(ROOT, "oracle", "_ref", "smalt"), os.path.join(ROOT, "oracle", "_ref", "smalt_gpu")
    gthreads = a.gpu_threads or a.threads
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        prefix = os.path.join(tmp, "hs")
        tot = int(sop[-1])
        indexfile.write_sma(prefix, names, sop, packed.cpu().numpy())
        indexfile.write_smi_perfect(prefix, k, s, idx.cpu().numpy(), pos.cpu().numpy(), (tot + s - 1) // s - 1)
        del ref, packed, idx, pos
        torch.cuda.empty_cache()

        def write_fq(path, n):
            q = b"I" * rlen
            with open(path, "wb") as f:
                for i in range(n):
                    f.write(b"@r%d\n" % i + rd[i].tobytes() + b"\n+\n" + q + b"\n")
            if a.paired:
                with open(path + ".mates", "wb") as f:
                    for i in range(n):
                        f.write(b"@r%d\n" % i + md[i].tobytes() + b"\n+\n" + q + b"\n")

        windows = {}

        def keep_failure(binary, nthr, r):
            """the whole stderr of a failing program goes to a file that travels back (a sweep that only reads our stdout
            must not lose the reason)"""
            d = os.path.join(ROOT, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "bench_tool_failure_%s_n%d_%d.err" % (os.path.basename(binary), nthr, int(time.time())))
            with open(path, "wb") as f:
                f.write(("exit status %d; environment: %s\n" % (r.returncode, {k: v for k, v in os.environ.items() if k.startswith("SMALTGPU")})).encode())
                f.write(r.stderr)
            sys.stderr.write("[bench_tool] stderr of the failing run kept in %s\n" % path)

        def run(binary, nthr, fq, out, env=None):
            t = time.time()
            inputs = [fq, fq + ".mates"] if a.paired else [fq]
            r = subprocess.run([binary, "map"] + a.extra.split() + ["-n", str(nthr), "-O", "-r", "-1", "-f", "cigar"] + (["-i", "500"] if a.paired else []) + ["-o", out, prefix] + inputs,
                               capture_output=True, env=env)
            if r.returncode:
                keep_failure(binary, nthr, r)
                raise SystemExit("%s failed: %s" % (binary, r.stderr.decode()[-1500:]))
            tm = [ln for ln in r.stderr.decode().split("\n") if "smaltgpu timing" in ln]      # the binding's per-phase seconds (integration/rmap_gpu.c)
            for ln in tm:
                if "window" in ln:
                    windows[fq] = float(ln.split()[3])
            if os.environ.get("SMALTGPU_TIMING"):
                sys.stderr.write("".join("[%s, %s] %s\n" % (os.path.basename(binary), os.path.basename(fq), ln) for ln in tm))
            return time.time() - t
        small, cpu_fq, gpu_fq = os.path.join(tmp, "s.fq"), os.path.join(tmp, "c.fq"), os.path.join(tmp, "g.fq")
        nsmall = min(2000, max(10, a.cpu_reads // 10))
        write_fq(small, nsmall)
        write_fq(cpu_fq, a.cpu_reads)
        write_fq(gpu_fq, a.reads)
        env = dict(os.environ, SMALTGPU_INDEX_PREFIX=prefix, SMALTGPU_TIMING="1")
        t_c0 = min(run(smalt, a.threads, small, os.path.join(tmp, "c0.cig")) for _ in range(2))
        t_c1 = run(smalt, a.threads, cpu_fq, os.path.join(tmp, "c1.cig"))
        t_g0 = min(run(smalt_gpu, gthreads, small, os.path.join(tmp, "g0.cig"), env) for _ in range(3))     # start-up (index load, device init) varies: best of three
        t_g1 = run(smalt_gpu, gthreads, gpu_fq, os.path.join(tmp, "g1.cig"), env)
        native = None
        if a.paired:
            # the same job by smaltgpu-map on the two files: libsmaltgpu only (ingest, rmapPair's rounds, pairing, report)
            prog = os.path.join(ROOT, "smalt_amd", "smaltgpu-map")
            subprocess.run([prog] + a.extra.split() + ["-r", "-1", "-i", "500", "-f", "cigar", "-n", str(a.threads), "-o", os.path.join(tmp, "n0.cig"), prefix, small, small + ".mates"], capture_output=True)
            r = subprocess.run([prog] + a.extra.split() + ["-r", "-1", "-i", "500", "-f", "cigar", "-n", str(a.threads), "-o", os.path.join(tmp, "n1.cig"), prefix, gpu_fq, gpu_fq + ".mates"],
                               capture_output=True, env=dict(os.environ, SMALTGPU_MAP_VERBOSE="1"))
            if r.returncode:
                keep_failure(prog, a.threads, r)
                raise SystemExit("smaltgpu-map failed: %s" % r.stderr.decode()[-1500:])
            ln = [x for x in r.stderr.decode().split("\n") if "lines out" in x][-1]
            sys.stderr.write("".join(x + "\n" for x in r.stderr.decode().split("\n") if x.startswith("smaltgpu-map:")))
            secs = float(ln.split("lines out")[1].split()[0])
            nat = open(os.path.join(tmp, "n1.cig")).read().split("\n")
            native = {"pairs_per_s": a.reads / secs, "window_s": secs, "index_load_s": float(ln.split("index load")[1].split()[0]), "host_threads": a.threads,
                      "identical_to_bound_program": nat == open(os.path.join(tmp, "g1.cig")).read().split("\n")}
        if not a.paired:
            # the same job by smaltgpu-map: libsmaltgpu only (read ingest, GPU path, post-processing, report) -- no reference code
            prog = os.path.join(ROOT, "smalt_amd", "smaltgpu-map")
            # once on the small file first: the first large device allocations after this script freed its 3 Gbp arrays are slow
            # (2 s per mapper, against 15 ms in a run of the program on its own: tools/run_native.sh)
            subprocess.run([prog] + a.extra.split() + ["-r", "-1", "-f", "cigar", "-n", str(a.threads), "-o", os.path.join(tmp, "n0.cig"), prefix, small], capture_output=True)
            runs = []
            for rep in range(max(1, a.native_repeat)):
                if a.native_gap > 0:
                    time.sleep(a.native_gap)
                r = subprocess.run([prog] + a.extra.split() + ["-r", "-1", "-f", "cigar", "-n", str(a.threads), "-o", os.path.join(tmp, "n1.cig"), prefix, gpu_fq], capture_output=True,
                                   env=dict(os.environ, SMALTGPU_MAP_VERBOSE="1"))
                if r.returncode:
                    keep_failure(prog, a.threads, r)
                    raise SystemExit("smaltgpu-map failed: %s" % r.stderr.decode()[-1500:])
                ln = [x for x in r.stderr.decode().split("\n") if "reads in to lines out" in x][-1]
                sys.stderr.write("".join(x + "\n" for x in r.stderr.decode().split("\n") if x.startswith("smaltgpu-map:")))
                runs.append(float(ln.split("lines out")[1].split()[0]))
            secs = min(runs)
            nat = open(os.path.join(tmp, "n1.cig")).read().split("\n")
            native = {"reads_per_s": a.reads / secs, "window_s": secs, "runs_window_s": runs, "reads_per_s_runs": [a.reads / x for x in runs], "index_load_s": float(ln.split("index load")[1].split()[0]), "host_threads": a.threads,
                      "identical_to_bound_program": nat == open(os.path.join(tmp, "g1.cig")).read().split("\n")}
        # same lines for the reads both programs mapped (the CPU run covers a prefix of the GPU run's reads)
        c = open(os.path.join(tmp, "c1.cig")).read().split("\n")
        g = open(os.path.join(tmp, "g1.cig")).read().split("\n")
        ncmp = len([x for x in c if x])
        identical = c[:ncmp] == g[:ncmp]
        cpu_rate = (a.cpu_reads - nsmall) / max(t_c1 - t_c0, 1e-6)
        gpu_rate = (a.reads - nsmall) / max(t_g1 - t_g0, 1e-6)
        print(json.dumps({"what": "whole program `smalt map`, %s/s with the index load removed" % ("read pairs" if a.paired else "reads"), "paired": a.paired, "extra_options": a.extra, "threads_cpu": a.threads, "threads_gpu_bound": gthreads,
                          "cpu_reads_per_s": cpu_rate, "gpu_bound_reads_per_s": gpu_rate, "speedup": gpu_rate / cpu_rate,
                          # the binding's own clock from "index resident" to exit (FASTQ input, mapping, output): free of the start-up noise
                          # that the difference of two program runs above carries
                          "gpu_bound_window_s": windows.get(gpu_fq), "gpu_bound_reads_per_s_window": (a.reads / windows[gpu_fq]) if windows.get(gpu_fq) else None,
                          "outputs_identical_on_common_reads": identical, "lines_compared": ncmp, "native_program": native,
                          "wall_s": {"cpu_small": t_c0, "cpu": t_c1, "gpu_small": t_g0, "gpu": t_g1}, "reads": {"cpu": a.cpu_reads, "gpu": a.reads}}))


if __name__ == "__main__":
    main()
