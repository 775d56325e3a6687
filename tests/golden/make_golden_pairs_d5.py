#!/usr/bin/env python3
"""A paired fixture for the mapping rounds without the best-only flag (`smalt map -d 5`): the round over the on-the-fly k = 5
index then keeps alignments down to the second-best first-pass score (rmap.c:1385-1400), e.g. two 7-base matches beside a
200-base alignment -- which change the first mate's share in the pair probabilities and its printed mapping quality (found by
tools/fuzz_pairs.py, case 25 of seed 2: the library printed 60 where `smalt map` prints 59).  Inputs: one chromosome and 24 pairs
of that case; expected: the dump of `oracle/_ref/refdump -P` (the reference's own rmapPair).  Adds its entry to
manifest_pairs.json.  Runs only in the build container (needs `make -C oracle ref`)."""
import gzip
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
TAG, K, S, OPTS, FIRST, NPAIRS = "gp_k13s2_d5", 13, 2, "-d 5 -i 1000 -j 27 -l pp", 418, 24


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, FUZZ_ONLY="25", FUZZ_KEEP=os.path.join(tmp, "case"))
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_pairs.py"), "48", "3000", "2"], check=True, env=env, capture_output=True)
        case = env["FUZZ_KEEP"]
        seqs, name = {}, None
        for ln in open(os.path.join(case, "ref.fa")):
            if ln.startswith(">"):
                name = ln[1:].split()[0]
                seqs[name] = []
            else:
                seqs[name].append(ln.strip())
        s = "".join(seqs["chr4"])
        fa = os.path.join(tmp, TAG + ".fa")
        with open(fa, "w") as f:
            f.write(">chr4\n")
            for o in range(0, len(s), 70):
                f.write(s[o:o + 70] + "\n")
        fqs = []
        for w in (1, 2):
            recs = open(os.path.join(case, "reads_%d.fq" % w)).read().split("\n")
            fq = os.path.join(tmp, "%s_%d.fq" % (TAG, w))
            with open(fq, "w") as f:
                for i in range(FIRST, FIRST + NPAIRS):
                    f.write("\n".join(recs[4 * i:4 * i + 4]) + "\n")
            fqs.append(fq)
        pre = os.path.join(tmp, TAG)
        subprocess.run([os.path.join(REF, "smalt"), "index", "-k", str(K), "-s", str(S), pre, fa], check=True, capture_output=True)
        dump = subprocess.run([os.path.join(REF, "refdump"), "-n", "-P", fqs[1]] + OPTS.split() + [pre, fqs[0]], check=True, capture_output=True).stdout
        for src, dst in ((fa, TAG + ".fa.gz"), (fqs[0], TAG + "_1.fq.gz"), (fqs[1], TAG + "_2.fq.gz")):
            with gzip.GzipFile(os.path.join(HERE, dst), "wb", mtime=0) as g:
                g.write(open(src, "rb").read())
        with gzip.GzipFile(os.path.join(HERE, TAG + ".refdump.txt.gz"), "wb", mtime=0) as g:
            g.write(dump)
    text = dump.decode()
    entry = dict(tag=TAG, k=K, s=S, opts=OPTS, npairs=NPAIRS, calls=text.count("\nMS ") + text.startswith("MS "),
                 restricted_calls=text.count(" fine=0") - text.count("niv=-1 fine=0"), fine_calls=text.count(" fine=1"), dump_lines=dump.count(b"\n"))
    mpath = os.path.join(HERE, "manifest_pairs.json")
    manifest = [e for e in json.load(open(mpath)) if e["tag"] != TAG] + [entry]
    json.dump(manifest, open(mpath, "w"), indent=1)
    print(entry)


if __name__ == "__main__":
    main()
