"""The reference's second embedded known-answer test, test/xali_test.py: reads that align across the junction of
consecutive reference sequences in concatenated mode (forced by -DSMALT_DEBUG_XALI, smalt.c:62-68), expected (reference
number, position, extended CIGAR, NM).  tests/golden/kat_xali.json holds the data of that script (lines 37-69).

CPU: the unmodified reference built with that macro (oracle/_ref/smalt_xali) reproduces the tuples -- this pins the
fixture and the build recipe.  GPU: the same program with its mapping worker bound to libsmaltgpu
(oracle/_ref/smalt_gpu_xali) prints the same records: the concatenated-mode path of the library (hit collection over the
whole set, alignments with sidx < 0 and concatenated offsets) feeding the reference's own assignSequenceIndex."""
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SMALT_XALI = os.path.join(ROOT, "oracle", "_ref", "smalt_xali")
SMALT_GPU_XALI = os.path.join(ROOT, "oracle", "_ref", "smalt_gpu_xali")
COMP = {"a": "t", "c": "g", "g": "c", "t": "a"}


def _files(tmp):
    doc = json.load(open(os.path.join(HERE, "golden", "kat_xali.json")))
    fa, rd, rc = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fa"), os.path.join(tmp, "reads_rc.fa")
    open(fa, "w").write("".join(">REF_%d\n%s\n" % (i + 1, s) for i, s in enumerate(doc["refseq"])))
    open(rd, "w").write("".join(">READ_%d\n%s\n" % (i + 1, r["seq"]) for i, r in enumerate(doc["reads"])))
    open(rc, "w").write("".join(">READ_%d\n%s\n" % (i + 1, "".join(COMP[c] for c in reversed(r["seq"]))) for i, r in enumerate(doc["reads"])))
    return doc, fa, rd, rc


def _records(path):
    out = []
    for ln in open(path):
        if ln.startswith("@"):
            continue
        f = ln.rstrip("\n").split("\t")
        nm = [int(t[5:]) for t in f[11:] if t.startswith("NM:i:")]
        out.append((f[0], int(f[2].rsplit("_", 1)[1]), int(f[3]), f[5], nm[0]))
    return out


def _check(prog, tmp, env=None):
    doc, fa, rd, rc = _files(tmp)
    pre = os.path.join(tmp, "idx")
    subprocess.run([SMALT_XALI, "index", "-k", str(doc["k"]), "-s", str(doc["s"]), pre, fa], check=True, capture_output=True)
    e = dict(os.environ, SMALTGPU_INDEX_PREFIX=pre, **(env or {}))
    sam = os.path.join(tmp, "out.sam")
    r = subprocess.run([prog, "map", "-f", "sam:x", "-o", sam, pre, rd], capture_output=True, env=e)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = _records(sam)
    assert len(got) == len(doc["reads"])
    for g, x in zip(got, doc["reads"]):
        assert g[1:] == (x["refno"], x["pos"], x["cigar_x"], x["nm"]), (g, x)
    sam_rc = os.path.join(tmp, "out_rc.sam")
    r = subprocess.run([prog, "map", "-f", "sam:x", "-o", sam_rc, pre, rc], capture_output=True, env=e)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return [ln for ln in open(sam) if not ln.startswith("@PG")], [ln for ln in open(sam_rc) if not ln.startswith("@PG")]


@pytest.mark.skipif(not os.path.exists(SMALT_XALI), reason="reference binary not built (make -C oracle ref)")
def test_reference_reproduces_the_xali_kat(tmp_path):
    _check(SMALT_XALI, str(tmp_path))


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(SMALT_XALI) and os.path.exists(SMALT_GPU_XALI)), reason="reference binaries not built (make -C oracle ref ref_gpu)")
@pytest.mark.parametrize("env", [{}, {"SMALTGPU_PER_READ": "1"}])
def test_bound_program_reproduces_the_xali_kat(env, tmp_path):
    d1, d2 = os.path.join(str(tmp_path), "a"), os.path.join(str(tmp_path), "b")
    os.makedirs(d1)
    os.makedirs(d2)
    fwd_ref, rc_ref = _check(SMALT_XALI, d1)
    fwd_gpu, rc_gpu = _check(SMALT_GPU_XALI, d2, env)
    assert fwd_gpu == fwd_ref
    assert rc_gpu == rc_ref            # the reverse-complemented reads (second half of the reference's script): same lines as the reference
