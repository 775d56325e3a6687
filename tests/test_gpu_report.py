"""`smaltgpu-map` (smalt_amd/csrc/smaltgpu_map_main.cpp), the file-to-file program on top of the C ABI -- reads parsed, mapped on
the GPU, post-processed, formatted, all by libsmaltgpu -- must print what the reference program `smalt map` printed for the same
command line: the committed `<tag>.<variant>.out.gz` files (tests/golden/make_golden_report.py; every case of
manifest_report.json, including those whose options change the mapping itself).  SURVEY 8f N4 + N1 + the hot path, end to end."""
import gzip
import json
import os
import subprocess

import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROG = os.path.join(ROOT, "smalt_amd", "smaltgpu-map")
REPORT_ALL = json.load(open(os.path.join(gu.GOLD, "manifest_report.json")))


@pytest.fixture(scope="module")
def fixtures(oracle_built, tmp_path_factory):
    tmp = tmp_path_factory.mktemp("rep")
    return {e["tag"]: gu.unpack(e, tmp) for e in gu.MANIFEST_ALL}, tmp


@pytest.mark.parametrize("case", REPORT_ALL, ids=["%s-%s" % (c["tag"], c["variant"]) for c in REPORT_ALL])
def test_program_prints_what_smalt_map_prints(case, fixtures):
    fxs, tmp = fixtures
    fx = fxs[case["tag"]]
    inp = gu.reshape_reads(fx["fq"], case["input"], str(tmp / ("%s.%s.txt" % (case["tag"], case["input"])))) if case.get("input") else fx["fq"]
    out = str(tmp / "out.txt")
    # small batches and several host threads: the blocks must come out in input order
    r = subprocess.run([PROG] + case["opts"] + ["-B", "64", "-n", "3", "-o", out, fx["prefix"], inp], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    with gzip.open(os.path.join(gu.GOLD, "%s.%s.out.gz" % (case["tag"], case["variant"])), "rb") as g:
        exp = g.read()
    got = open(out, "rb").read()
    gl = [x for x in got.split(b"\n") if not x.startswith(b"@PG")]
    el = [x for x in exp.split(b"\n") if not x.startswith(b"@PG")]
    assert len(gl) == len(el)
    for i, (x, y) in enumerate(zip(gl, el)):
        assert x == y, (i, x, y)


def test_program_large_batch_equals_small_batches(fixtures):
    fxs, tmp = fixtures
    fx = fxs["g_k13s6_ties"]
    outs = []
    # "-g 0,0": two images of the index (the second a device-to-device copy), four mappers -- the N-device path on one GPU
    for extra in (["-B", "50", "-n", "2"], ["-n", "8"], ["-B", "20", "-g", "0,0"]):
        out = str(tmp / "cmp.txt")
        r = subprocess.run([PROG, "-r", "11", "-f", "sam:nohead"] + extra + ["-o", out, fx["prefix"], fx["fq"]], capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] == outs[2] and outs[0].count(b"\n") >= 220


def test_program_reads_gzip_input(fixtures):
    """the reference reads gzip-compressed FASTQ through zlib (sequence.c:1108); smaltgpu-map inflates the stream itself"""
    fxs, tmp = fixtures
    fx = fxs["g_k13s6_hash"]
    gz = str(tmp / "reads.fq.gz")
    raw = open(fx["fq"], "rb").read()
    half = len(raw) // 2
    half = raw.index(b"\n@r", half) + 1
    with open(gz, "wb") as f:                       # two gzip members, as `cat a.gz b.gz` gives
        f.write(gzip.compress(raw[:half]) + gzip.compress(raw[half:]))
    outs = []
    for inp in (fx["fq"], gz):
        out = str(tmp / "gz.txt")
        r = subprocess.run([PROG, "-r", "3", "-f", "sam:nohead", "-B", "100", "-o", out, fx["prefix"], inp], capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and outs[0].count(b"\n") == 250


def test_program_with_the_anti_diagonal_band_form(fixtures):
    """SMALTGPU_ALIGN_ANTIDIAG=1 keeps K3's older form for narrow bands (band_track_wave + the one-lane traceback) reachable:
    the program must print the same lines with it (the row form is the default everywhere else)."""
    fxs, tmp = fixtures
    for tag, variant in (("g_k13s6_ties", "sam"), ("g_k11s2_d20", "cigar")):
        case = [c for c in REPORT_ALL if c["tag"] == tag and c["variant"] == variant][0]
        out = str(tmp / "anti.txt")
        r = subprocess.run([PROG] + case["opts"] + ["-o", out, fxs[tag]["prefix"], fxs[tag]["fq"]], capture_output=True,
                           env=dict(os.environ, SMALTGPU_ALIGN_ANTIDIAG="1"))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        with gzip.open(os.path.join(gu.GOLD, "%s.%s.out.gz" % (tag, variant)), "rb") as g:
            exp = [x for x in g.read().split(b"\n") if not x.startswith(b"@PG")]
        got = [x for x in open(out, "rb").read().split(b"\n") if not x.startswith(b"@PG")]
        assert got == exp
