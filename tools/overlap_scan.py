"""Statement-level overlap of our sources with the reference tree (the check the review applies): comments and white
space stripped, text split at ; { }, statements of 14 characters and more that occur verbatim in /root/reference/src/*.c
are counted.  Usage: python tools/overlap_scan.py [files...] (default: every product and integration source)."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def statements(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"\s+", "", text)
    return [s for s in re.split(r"[;{}]", text) if len(s) >= 14]


def main():
    ref = set()
    for f in glob.glob("/root/reference/src/*.[ch]"):
        ref.update(statements(open(f, errors="replace").read()))
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "smalt_amd/csrc/*.[ch]*")) + glob.glob(os.path.join(ROOT, "integration/*.[ch]")))
    for f in files:
        if f.endswith(".o"):
            continue
        st = statements(open(f, errors="replace").read())
        hit = [s for s in st if s in ref]
        print("%-48s %4d / %4d = %4.1f %%" % (os.path.relpath(f, ROOT), len(hit), len(st), 100.0 * len(hit) / max(1, len(st))))
        if os.environ.get("SHOW"):
            for s in hit:
                print("      ", s[:120])


if __name__ == "__main__":
    main()
