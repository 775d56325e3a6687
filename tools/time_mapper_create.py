#!/usr/bin/env python3
"""Diagnostic: seconds to create (and free) a mapper for 262144 x 150-base reads at several scratch budgets."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from smalt_amd import api, gpuindex
    dev = torch.device("cuda", 0)
    nchr, chrlen = 4, 1000000
    sop = np.arange(nchr + 1, dtype=np.int64) * chrlen
    ref = gpuindex.make_reference_gpu(nchr, chrlen, 1, dev)
    asc = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[ref.long()]
    torch.cuda.synchronize()
    gix = api.Index.build_device(asc.data_ptr(), [int(x) for x in sop], ["c%d" % i for i in range(nchr)], 13, 6, 0)
    for budget in (0, 28, 8, 2):
        for rep in range(2):
            t = time.time()
            mp = api.Mapper(gix, 262144, 150, slot_budget_gb=budget)
            t1 = time.time() - t
            t = time.time()
            mp.close()
            print("budget %2d GB: create %.3f s, free %.3f s" % (budget, t1, time.time() - t), flush=True)
    gix.close()


if __name__ == "__main__":
    main()
