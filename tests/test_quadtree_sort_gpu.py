"""The careful loop's std::sort (ORBextractor.cc:700) on the device against libstdc++ itself.

std::sort is unstable: the order of nodes with equal (count, UL.x) is whatever libstdc++'s introsort leaves, and it decides which
nodes DistributeOctTree splits before the quota is reached.  The selection kernels restate the algorithm (quadtree_device.h) in two
device forms — the frame form (1024 threads: a queue of ranges, ranges of <= 64 items finished in the lanes of one wave by
wave_introsort64) and the batch form (256 threads, level-synchronous rounds).  msorb_debug_std_sort runs them alone on explicit keys;
the oracle runs std::sort (oracle/orb_extractor_oracle.cc orc_std_sort_order).  Tie-heavy inputs of every size class."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(11)
    out = []
    sizes = list(range(0, 40)) + [47, 48, 49, 63, 64, 65, 66, 80, 97, 127, 128, 129, 150, 200, 255, 256, 257, 300, 434, 500, 777, 1000, 2000]
    for n in sizes:
        for kind in range(6):
            if kind == 0:   # (count, x0) keys of a careful sweep: few distinct counts, column boundaries
                k = (rng.integers(2, 12, n).astype(np.uint32) << 16) | rng.choice(np.array([0, 77, 155, 232, 310, 387, 465, 542, 620], np.uint32), n)
            elif kind == 1:  # two values only
                k = rng.integers(0, 2, n).astype(np.uint32)
            elif kind == 2:  # all equal
                k = np.full(n, 7, np.uint32)
            elif kind == 3:  # already sorted / reversed with ties
                k = np.sort(rng.integers(0, max(2, n // 3), n).astype(np.uint32))
                if n % 2:
                    k = k[::-1].copy()
            elif kind == 4:  # distinct
                k = rng.permutation(n).astype(np.uint32)
            else:            # organ pipe: a killer pattern for median-of-three (deep recursion, the depth budget)
                h = np.arange(n // 2, dtype=np.uint32)
                k = np.concatenate([h, h[::-1], np.zeros(n - 2 * (n // 2), np.uint32)])
            out.append(k)
    return out


@pytest.mark.parametrize("frame_form", [True, False])
def test_device_std_sort_is_libstdcxx_std_sort(msorb_mod, oracle, frame_form):
    bad = []
    for k in _cases():
        order, sk = msorb_mod.debug_std_sort(k, frame_form=frame_form)
        want = oracle.std_sort_order(k)
        if not (np.array_equal(order, want) and np.array_equal(sk, k[want])):
            bad.append((len(k), int(np.unique(k).size)))
    assert not bad, f"{len(bad)} key sequences sorted into another permutation than std::sort's (n, distinct keys): {bad[:12]}"


def test_device_std_sort_random_campaign(msorb_mod, oracle):
    """2 000 random tie-heavy sequences around the sizes where the frame form changes its method (<= 16: one leaf; <= 64: the lanes of
    one wave; above: LDS partitions that hand ranges of <= 64 items to the lanes)."""
    rng = np.random.default_rng(5)
    for it in range(2000):
        n = int(rng.choice([rng.integers(0, 20), rng.integers(17, 66), rng.integers(60, 140), rng.integers(100, 450)]))
        distinct = int(rng.choice([1, 2, 3, 5, 8, 20, 100, 1 << 20]))
        k = rng.integers(0, distinct, n).astype(np.uint32)
        order, _ = msorb_mod.debug_std_sort(k, frame_form=True)
        assert np.array_equal(order, oracle.std_sort_order(k)), (it, n, distinct)
