"""N > 1 path on CPU: world_size 2, backend gloo.  Rank 0 builds the index image and broadcasts it;
both ranks map their shard of the reads (with the oracle standing in for the device -- the product
path itself has no CPU form) and the rank-ordered concatenation must equal mapping all reads in one
process.  Also the bench.py reductions (max time, summed counters)."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def test_shard_bounds_tile_the_input():
    from smalt_amd.shard import shard_bounds
    for n in (0, 1, 2, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, ret):
    import torch
    import torch.distributed as dist
    import oracle_lib as ol
    from smalt_amd import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    k, s = 9, 3
    ch = synth.make_reference(2, 200000, seed=5, repeat_frac=0.1, n_fam=2, cons_len=200)
    reads, _ = synth.make_reads(ch, 41, 80, seed=6, sub_rate=0.02, indel_read_frac=0.2)     # 41: uneven shards
    rb = [synth.codes_to_ascii(r) for r in reads]
    seqs = [synth.codes_to_ascii(c) for c in ch]
    image = {}
    if rank == 0:                              # only rank 0 builds the index image
        oix = ol.build_index(seqs, ["c0", "c1"], k, s)
        o = oix.contents
        assert o.typ == 0
        image = {"idx": torch.from_numpy(np.ctypeslib.as_array(o.idx, shape=(o.nkeys + 1,)).astype(np.int32)),
                 "pos": torch.from_numpy(np.ctypeslib.as_array(o.pos, shape=(o.npos,)).astype(np.int32)),
                 "packed": torch.from_numpy(np.ctypeslib.as_array(o.packed, shape=(o.totlen // 10 + 1,)).astype(np.int32))}
    got, secs = shard.broadcast_image(image, dev, 0, order=("idx", "pos", "packed"))
    assert secs >= 0
    sums = [int(got[n].to(torch.int64).sum().item()) for n in ("idx", "pos", "packed")]
    all_sums = shard.gather_in_rank_order(sums)
    assert all(x == all_sums[0] for x in all_sums)              # every rank holds the same image
    # every rank rebuilds its mapper from the broadcast image (here: writes the index files the oracle reads)
    from smalt_amd import indexfile
    pre = os.path.join(tmp, "ix%d" % rank)
    sop = np.array([0, 200000, 400000], dtype=np.int64)
    indexfile.write_sma(pre, ["c0", "c1"], sop, got["packed"].numpy().view(np.uint32))
    idx_u, pos_u = got["idx"].numpy().view(np.uint32), got["pos"].numpy().view(np.uint32)
    indexfile.write_smi_perfect(pre, k, s, idx_u, pos_u, int(pos_u.max()) if pos_u.size else 0)
    oix2 = ol.lib().or_index_read(pre.encode())
    assert oix2
    lo, hi = shard.shard_bounds(len(rb), rank, world)
    m = ol.Mapper(oix2)
    par = ol.default_params(oix2)
    mine = []
    for r in rb[lo:hi]:
        rv, res = m.map(r, b"I" * len(r), par)
        assert rv == 0
        mine.append(res)
    m.close()
    parts = shard.gather_in_rank_order(mine)
    merged = [x for p in parts for x in p]
    dt, (nmapped, nreads) = shard.reduce_step(1.0 + rank, [sum(1 for x in mine if x), len(mine)], dev)
    assert dt == float(world) and nreads == len(rb)
    if rank == 0:
        m = ol.Mapper(oix)
        exp = []
        for r in rb:
            rv, res = m.map(r, b"I" * len(r), par)
            exp.append(res)
        m.close()
        assert merged == exp
        assert nmapped == sum(1 for x in exp if x)
        ret.put("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_mapping_matches_single_process(oracle_built, tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ret.get(timeout=5) == "ok"


def test_bench_starts_its_own_ranks_and_deals_sub_batches():
    """`python bench.py --gpus 2` (no launcher) must itself produce two ranks; the shared cursor deals every sub-batch
    exactly once.  Dry run: stops before the device is needed."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, SMALT_BENCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    for extra in ([], ["--static-shards"]):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + extra, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["backend"] == "gloo"
        a, b = line["dealt"]
        assert sorted(a + b) == list(range(23)) and a and b
        if extra:
            assert a == list(range(12)) and b == list(range(12, 23))
        # guided dealing (the N > 1 job): the pieces of both ranks tile the job's reads exactly once, and they get smaller
        # towards the end
        ga, gb = line["guided"]
        allp = sorted([tuple(x) for x in ga + gb])
        assert allp[0][0] == 0 and sum(c for _, c in allp) == 1000003 and ga and gb
        assert all(allp[i][0] + allp[i][1] == allp[i + 1][0] for i in range(len(allp) - 1))
        if not extra:
            assert max(c for _, c in allp) == 65536 and allp[-1][1] <= 8192
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")], env=env, capture_output=True, text=True, timeout=300)
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["dealt"] == [list(range(23))]
