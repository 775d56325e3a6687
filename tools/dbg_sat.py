"""Debug helper (GPU box): the satellite-repeat pairs of tests/test_gpu_dropin.py through map_batch, read by read."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_dropin as t
from smalt_amd import api
tmp = tempfile.mkdtemp()
pre, fqs, pairs = t.satellite_pairs(tmp)
ix = api.Index.load(pre)
par = ix.default_params()
for cap in (256, 4096):
    m = api.Mapper(ix, cap, 256)
    for w in (0, 1):
        reads = [p[w] for p in pairs[:14]]
        res, st = m.map_batch(reads, None, par, allow_read_errors=True)
        print("cap", cap, "mate", w + 1, [(s["err"], s["nseg"], s["nhit"]) for s in st], flush=True)
        res, st = m.map_batch(reads[:2], None, par, allow_read_errors=True)
        print("  two reads:", [(s["err"], s["nseg"], s["nhit"]) for s in st], flush=True)
    m.close()
