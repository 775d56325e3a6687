import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_dropin as t
tmp = tempfile.mkdtemp()
pre, fqs, pairs = t.satellite_pairs(tmp, copies=1300)
env = dict(os.environ, SMALTGPU_INDEX_PREFIX=pre, SMALTGPU_TIMING="1")
r = subprocess.run([t.SMALT_GPU, "map", "-f", "cigar", "-i", "500", "-r", "-1", "-x", "-o", os.path.join(tmp, "o"), pre] + fqs, capture_output=True, env=env)
print([ln for ln in r.stderr.decode().split("\n") if "interval" in ln])
