"""Wave-parallel candidate ranking sort (smg_wsort.hpp) against the oracle's restatement of the
reference's unstable quicksort (sort.c:233): the permutation -- i.e. the order of equal keys --
must be identical, in LDS and in HBM, with and without a rank limit."""
import os
import tempfile

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _arrays(rng):
    arrs = []
    for n in list(range(0, 80)) + [100, 127, 128, 129, 255, 256, 257, 500, 1000, 1080, 2047, 2048, 4097, 8192, 9000, 20000]:
        for kmax in (1, 2, 3, 8, 40, 150, 1000):
            arrs.append(rng.integers(0, kmax, size=n, dtype=np.uint32))
    for n in (64, 300, 1500, 7000):         # sorted, reversed, organ pipe, constant: worst cases of median-of-3
        a = np.sort(rng.integers(0, 100, size=n, dtype=np.uint32))
        arrs += [a, a[::-1].copy(), np.concatenate([a[::2], a[::-2]]), np.full(n, 7, dtype=np.uint32)]
    return arrs


def _expect(a):
    k = np.ascontiguousarray(a, dtype=np.uint32).copy()
    v = np.arange(len(a), dtype=np.uint32)
    if len(a):
        ol.lib().or_sort2_u32(len(a), k.ctypes.data, v.ctypes.data)
    return k, v


@pytest.fixture(scope="module")
def mapper(oracle_built):
    from smalt_amd import api
    rng = np.random.default_rng(5)
    oix = ol.build_index([bytes(rng.choice(list(b"ACGT"), size=2000).astype(np.uint8))], ["s"], 11, 3)
    with tempfile.TemporaryDirectory() as tmp:
        pre = os.path.join(tmp, "x")
        ol.lib().or_index_write(oix, pre.encode())
        gix = api.Index.load(pre, 0)
    mp = api.Mapper(gix, 16, 128)
    yield mp
    mp.close()
    gix.close()
    ol.lib().or_index_free(oix)


@pytest.mark.parametrize("in_lds", [True, False])
def test_rank_sort_full(mapper, in_lds):
    arrs = _arrays(np.random.default_rng(11 + in_lds))
    got = mapper.rank_sort_batch(arrs, -1, in_lds)
    for a, (gk, gi) in zip(arrs, got):
        ek, ei = _expect(a)
        assert np.array_equal(gk, ek), len(a)
        assert np.array_equal(gi, ei), (len(a), int(a.max()) if len(a) else 0)


@pytest.mark.parametrize("nneed", [1, 33, 512, 700])
def test_rank_sort_prefix(mapper, nneed):
    arrs = _arrays(np.random.default_rng(100 + nneed))
    got = mapper.rank_sort_batch(arrs, nneed, True)
    for a, (gk, gi) in zip(arrs, got):
        ek, ei = _expect(a)
        m = min(nneed, len(a))
        assert np.array_equal(gi[:m], ei[:m]), (len(a), nneed)
        assert np.array_equal(np.sort(gi), np.arange(len(a)))        # still a permutation
