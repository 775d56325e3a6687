#!/usr/bin/env python3
"""Split-read fixtures `gs_*`: chimeric reads (tests/test_gpu_split.py: chimeric_reads) and what the reference's own rmapSingle
does with them under RMAPFLG_SPLIT (`smalt map -p`) -- `oracle/_ref/refdump -s -p -n`: one block per mapSingleRead call (the read's
own call, then mapSecondary's with k-mer words from the stretch the best alignment leaves uncovered, rmap.c:1435-1505), each with
its arguments (`MS`), stage state, the alignments it added (`RS`, `RX`) and the set as resultSetSortAndAssignSequence leaves it
(`PS`, `RF`, `SO`, `SS`, `SG`); behind the `PE` line the set as rmapSingle returns it; and what `smalt map -p -r -1` prints for the
reads (`<tag>.split_cigar.out.gz`, `.split_sam.out.gz`).  Writes manifest_split.json.
Data only; runs in the build container (needs `make -C oracle ref`), not run by the tests.

    python tests/golden/make_golden_split.py
"""
import gzip
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.path.join(ROOT, "oracle", "_ref")
# tag, k, s, sequences, their length, reads, read length, refdump options, seed
CASES = [("gs_k13s4", 13, 4, 3, 60_000, 120, 150, "", 811),
         ("gs_k11s3_q10", 11, 3, 5, 40_000, 90, 120, "-q 10", 812)]


def main():
    import test_gpu_split as tgs
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    man = []
    with tempfile.TemporaryDirectory() as tmp:
        for tag, k, s, nchr, chrlen, nreads, rlen, opts, seed in CASES:
            d = os.path.join(tmp, tag)
            os.mkdir(d)
            _, fa, fq, _ = tgs.write_inputs(d, nchr, chrlen, nreads, rlen, seed)
            pre = os.path.join(d, "ix")
            subprocess.run([os.path.join(REF, "smalt"), "index", "-k", str(k), "-s", str(s), pre, fa], check=True, capture_output=True)
            dump = subprocess.run([os.path.join(REF, "refdump"), "-s", "-p", "-n"] + opts.split() + [pre, fq], check=True, capture_output=True).stdout
            for src, dst in ((fa, tag + ".fa.gz"), (fq, tag + ".fq.gz")):
                with gzip.GzipFile(os.path.join(HERE, dst), "wb", mtime=0) as g:
                    g.write(open(src, "rb").read())
            with gzip.GzipFile(os.path.join(HERE, tag + ".refdump.txt.gz"), "wb", mtime=0) as g:
                g.write(dump)
            # what the reference program prints for the same reads (tests/test_split_report.py: the report of split reads on the CPU)
            for fmt in ("cigar", "sam"):
                out = os.path.join(d, "o." + fmt)
                subprocess.run([os.path.join(REF, "smalt"), "map", "-p", "-r", "-1", "-f", fmt] + opts.split() + ["-o", out, pre, fq], check=True, capture_output=True)
                with gzip.GzipFile(os.path.join(HERE, "%s.split_%s.out.gz" % (tag, fmt)), "wb", mtime=0) as g:
                    g.write(open(out, "rb").read())
            text = dump.decode()
            e = dict(tag=tag, k=k, s=s, opts=opts, nreads=nreads, calls=text.count("\nMS "), second_calls=text.count("\nMS 1 "), dump_lines=dump.count(b"\n"))
            man.append(e)
            print(e)
    json.dump(man, open(os.path.join(HERE, "manifest_split.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
