"""Helpers shared by the golden-fixture tests (CPU and GPU): unpack a committed fixture,
build its index with the oracle's own writer and return paths + expected dump text."""
import gzip
import hashlib
import json
import os

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
MANIFEST_ALL = json.load(open(os.path.join(GOLD, "manifest.json")))
MANIFEST = [e for e in MANIFEST_ALL if not e.get("post_only")]        # fixtures whose stage dumps the path reproduces line by line


def read_fasta(path):
    names, seqs = [], []
    with open(path) as f:
        for ln in f:
            ln = ln.strip()
            if not ln:
                continue
            if ln[0] == ">":
                names.append(ln[1:].split()[0])
                seqs.append([])
            else:
                seqs[-1].append(ln)
    return names, ["".join(s).encode() for s in seqs]


def read_fastq(path):
    out = []
    with open(path, "rb") as f:
        while True:
            n = f.readline()
            if not n:
                break
            s = f.readline().rstrip(b"\r\n")
            f.readline()
            q = f.readline().rstrip(b"\r\n")
            out.append((n[1:].split()[0].decode(), s, q))
    return out


def unpack(entry, tmpdir):
    """-> dict(prefix, fa, fq, expected_dump:str); writes <prefix>.sma/.smi via the oracle."""
    tag = entry["tag"]
    paths = {}
    for ext in ("fa", "fq"):
        p = os.path.join(str(tmpdir), "%s.%s" % (tag, ext))
        with gzip.open(os.path.join(GOLD, "%s.%s.gz" % (tag, ext)), "rb") as g, open(p, "wb") as f:
            f.write(g.read())
        paths[ext] = p
    names, seqs = read_fasta(paths["fa"])
    ix = ol.build_index(seqs, names, entry["k"], entry["s"])
    prefix = os.path.join(str(tmpdir), tag)
    assert ol.lib().or_index_write(ix, prefix.encode()) == 0
    ol.lib().or_index_free(ix)
    with gzip.open(os.path.join(GOLD, tag + ".refdump.txt.gz"), "rt") as g:
        expected = g.read()
    return dict(prefix=prefix, fa=paths["fa"], fq=paths["fq"], expected=expected, names=names, seqs=seqs)


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def reshape_reads(fq_path, style, out_path):
    """The reads of a FASTQ fixture in another legal spelling (for the ingest tests):
    'wrapped' = FASTQ with header comments, tabs, sequence and quality lines wrapped at 60, lower-case bases, blank lines and
    CRLF line ends here and there; 'fasta' = FASTA wrapped at 70 without qualities."""
    recs = read_fastq(fq_path)
    with open(out_path, "wb") as f:
        for i, (nm, sq, ql) in enumerate(recs):
            eol = b"\r\n" if i % 3 == 1 else b"\n"
            if style == "fasta":
                f.write(b">" + nm.encode() + b" len=%d" % len(sq) + eol)
                for o in range(0, len(sq), 70):
                    f.write(sq[o:o + 70] + eol)
            else:
                f.write(b"@" + (b"  " if i % 4 == 2 else b"") + nm.encode() + b"\tcomment %d  x" % i + eol)
                body = sq.lower() if i % 5 == 0 else sq
                for o in range(0, len(sq), 60):
                    f.write(body[o:o + 60] + eol)
                f.write(b"+" + (nm.encode() if i % 2 else b"") + eol)
                for o in range(0, len(ql), 60):
                    f.write(ql[o:o + 60] + eol)
                if i % 7 == 3:
                    f.write(eol)
    return out_path
