"""A bounded run of the two fuzzers (tools/fuzz_single.py, tools/fuzz_pairs.py): randomised reference shapes, index word lengths,
read lengths, library types, output formats and search options; `smaltgpu-map` must print what the unmodified `smalt map` prints.
The long runs behind profiles/r03_fuzz_summary.txt used the same tools with more cases per seed."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALT = os.path.join(ROOT, "oracle", "_ref", "smalt")


@pytest.mark.skipif(not os.path.exists(SMALT), reason="reference binary not built (make -C oracle ref)")
@pytest.mark.parametrize("tool,ncases,n,seed", [("fuzz_single.py", 10, 1500, 21), ("fuzz_pairs.py", 10, 1200, 22)], ids=["single", "pairs"])
def test_random_configurations_print_what_the_reference_prints(tool, ncases, n, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(ncases), str(n), str(seed)], capture_output=True, text=True)
    ok = sum(1 for ln in r.stdout.split("\n") if " ok:" in ln)
    assert r.returncode == 0 and ok >= ncases - 3, r.stdout[-3000:] + r.stderr[-1000:]     # (a case the reference itself rejects is not counted)
