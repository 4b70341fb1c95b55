"""world_size-2 gloo test of the N>1 data path (stereo left/right split): the right-eye rank's feature
block must arrive bit-identical on the left-eye rank; ranks without a partner do nothing."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "ms-slam_amd"))
    os.environ["MSORB_NO_TORCH"] = "1"
    from msorb import stereo_split as ss
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, cap = 6, 2024
    mine, theirs = ss.FeatureBlock(n, cap, "cpu"), ss.FeatureBlock(n, cap, "cpu")
    g = torch.Generator().manual_seed(100 + rank)
    mine.counts.copy_(torch.randint(0, cap, (n,), generator=g, dtype=torch.int32))
    mine.kps.copy_(torch.randint(0, 256, mine.kps.shape, generator=g, dtype=torch.uint8))
    mine.desc.copy_(torch.randint(0, 256, mine.desc.shape, generator=g, dtype=torch.uint8))
    for step in range(3):
        has_both = ss.exchange(dist, rank, world, mine, theirs)
        dist.barrier()
    ok = True
    if ss.eye_of(rank) == 0 and ss.partner_of(rank, world) is not None:
        g2 = torch.Generator().manual_seed(100 + ss.partner_of(rank, world))
        ok = has_both and torch.equal(theirs.counts, torch.randint(0, cap, (n,), generator=g2, dtype=torch.int32))
        ok = ok and torch.equal(theirs.kps, torch.randint(0, 256, mine.kps.shape, generator=g2, dtype=torch.uint8))
        ok = ok and torch.equal(theirs.desc, torch.randint(0, 256, mine.desc.shape, generator=g2, dtype=torch.uint8))
    else:
        ok = not has_both
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _worker_async(rank, world, port, q):
    """bench.py's N>1 step loop: two blocks used alternately, exchange of step k in flight during step k+1."""
    sys.path.insert(0, os.path.join(ROOT, "ms-slam_amd"))
    os.environ["MSORB_NO_TORCH"] = "1"
    from msorb import stereo_split as ss
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, cap = 4, 300
    mine = [ss.FeatureBlock(n, cap, "cpu") for _ in range(2)]
    theirs = [ss.FeatureBlock(n, cap, "cpu") for _ in range(2)]
    pending = [[], []]
    ok = True

    def fill(blk, who, step):
        g = torch.Generator().manual_seed(1000 * who + step)
        blk.counts.copy_(torch.randint(0, cap, (n,), generator=g, dtype=torch.int32))
        blk.kps.copy_(torch.randint(0, 256, blk.kps.shape, generator=g, dtype=torch.uint8))
        blk.desc.copy_(torch.randint(0, 256, blk.desc.shape, generator=g, dtype=torch.uint8))

    def check(b, step):
        want = ss.FeatureBlock(n, cap, "cpu")
        fill(want, ss.partner_of(rank, world), step)
        return all(torch.equal(a, c) for a, c in zip(theirs[b].tensors(), want.tensors()))

    steps = 5
    for step in range(steps):
        b = step & 1
        ss.finish(pending[b])
        if pending[b] and ss.eye_of(rank) == 0:
            ok = ok and check(b, step - 2)       # what arrived two steps ago, before the block is reused
        pending[b] = []
        fill(mine[b], rank, step)                # "extraction" of this step into the free block
        pending[b] = ss.exchange_async(dist, rank, world, mine[b], theirs[b])
    for b in range(2):
        ss.finish(pending[b])
    if ss.eye_of(rank) == 0 and ss.partner_of(rank, world) is not None:
        ok = ok and check((steps - 1) & 1, steps - 1) and check((steps - 2) & 1, steps - 2)
    dist.barrier()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _worker_halves(rank, world, port, q):
    """bench.py's N>1 step loop with the association split over both ranks: every rank ships the upper half of its block and
    receives its partner's upper half (the pairs it holds first)."""
    sys.path.insert(0, os.path.join(ROOT, "ms-slam_amd"))
    os.environ["MSORB_NO_TORCH"] = "1"
    from msorb import stereo_split as ss
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, cap = 6, 200
    h = n // 2
    mine = [ss.FeatureBlock(n, cap, "cpu") for _ in range(2)]
    theirs = [ss.FeatureBlock(h, cap, "cpu") for _ in range(2)]
    pending = [[], []]
    ok = True

    def fill(blk, who, step):
        g = torch.Generator().manual_seed(1000 * who + step)
        blk.counts.copy_(torch.randint(0, cap, (n,), generator=g, dtype=torch.int32))
        blk.kps.copy_(torch.randint(0, 256, blk.kps.shape, generator=g, dtype=torch.uint8))
        blk.desc.copy_(torch.randint(0, 256, blk.desc.shape, generator=g, dtype=torch.uint8))

    def check(b, step):
        want = ss.FeatureBlock(n, cap, "cpu")
        fill(want, ss.partner_of(rank, world), step)
        return all(torch.equal(a, c[h:]) for a, c in zip(theirs[b].tensors(), want.tensors()))

    steps = 5
    has_partner = ss.partner_of(rank, world) is not None
    for step in range(steps):
        b = step & 1
        ss.finish(pending[b])
        if pending[b]:
            ok = ok and check(b, step - 2)
        pending[b] = []
        fill(mine[b], rank, step)
        pending[b] = ss.swap_halves_async(dist, rank, world, mine[b], theirs[b])
        ok = ok and (bool(pending[b]) == has_partner)
    for b in range(2):
        ss.finish(pending[b])
    if has_partner:
        ok = ok and check((steps - 1) & 1, steps - 1) and check((steps - 2) & 1, steps - 2)
    dist.barrier()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])   # 4 = two eye pairs: (0, 1) and (2, 3) swap independently
def test_stereo_split_swap_halves(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker_halves, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
    assert res == {r: True for r in range(world)}


@pytest.mark.parametrize("world", [2, 3])
def test_stereo_split_async_double_buffer(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker_async, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
    assert res == {r: True for r in range(world)}


@pytest.mark.parametrize("world", [2, 3])
def test_stereo_split_exchange(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
    assert res == {r: True for r in range(world)}


def test_pairing_rules():
    sys.path.insert(0, os.path.join(ROOT, "ms-slam_amd"))
    from msorb import stereo_split as ss
    assert [ss.eye_of(r) for r in range(4)] == [0, 1, 0, 1]
    assert [ss.partner_of(r, 4) for r in range(4)] == [1, 0, 3, 2]
    assert ss.partner_of(2, 3) is None and ss.pair_group(5) == 2
