"""2-GPU stereo split (BASELINE.json configs[3]): the left eye's ORBextractor lives on one GPU, the right
eye's on its partner; after extraction the right-eye rank sends its fixed-capacity feature block
(counts, keypoints, descriptors) to the left-eye rank over RCCL point-to-point (xGMI), which replaces the
join of the two extraction threads in Frame.cc:122-125.  No collective on the data path: one send/recv
pair per step.  Works with any torch.distributed backend (gloo on CPU for the tests)."""
import torch


def eye_of(rank):
    """0 = left-eye rank (receiver), 1 = right-eye rank (sender)."""
    return rank % 2


def partner_of(rank, world):
    p = rank ^ 1
    return p if p < world else None


def pair_group(rank):
    return rank // 2


class FeatureBlock:
    """Fixed-capacity per-step payload: counts[n] int32, keypoints[n,cap,28] u8, descriptors[n,cap,32] u8."""

    def __init__(self, n_images, capacity, device):
        self.counts = torch.zeros(n_images, dtype=torch.int32, device=device)
        self.kps = torch.zeros((n_images, capacity, 28), dtype=torch.uint8, device=device)
        self.desc = torch.zeros((n_images, capacity, 32), dtype=torch.uint8, device=device)

    def tensors(self):
        return (self.counts, self.kps, self.desc)

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.tensors())


def exchange(dist, rank, world, mine, theirs):
    """Right-eye ranks send `mine`; left-eye ranks receive into `theirs`.  Returns True if this rank now holds
    both eyes' features (left-eye rank with a partner)."""
    p = partner_of(rank, world)
    if p is None:
        return False
    if eye_of(rank) == 1:
        for t in mine.tensors():
            dist.send(t, p)
        return False
    for t in theirs.tensors():
        dist.recv(t, p)
    return True


def exchange_async(dist, rank, world, mine, theirs):
    """Non-blocking form of exchange(): returns the list of Work handles (empty without a partner).  The caller must
    not touch `mine` (sender) / `theirs` (receiver) until finish() has returned for these handles — the extraction
    kernels run on the library's own HIP streams, which torch does not order against its communication stream."""
    p = partner_of(rank, world)
    if p is None:
        return []
    if eye_of(rank) == 1:
        return [dist.isend(t, p) for t in mine.tensors()]
    return [dist.irecv(t, p) for t in theirs.tensors()]


def swap_halves_async(dist, rank, world, mine, theirs):
    """Symmetric form for load balance: both ranks of a pair hold 2h images of their eye, ordered so that each rank
    associates the h stereo pairs of its FIRST h images and ships the features of its LAST h images to the partner (who holds
    those pairs first).  `mine` has 2h images, `theirs` h.  Non-blocking; returns the Work handles (empty without a partner);
    finish() before touching either block."""
    p = partner_of(rank, world)
    if p is None:
        return []
    h = theirs.counts.shape[0]
    sends = [t[h:] for t in mine.tensors()]
    recvs = list(theirs.tensors())
    if hasattr(dist, "batch_isend_irecv"):
        # one group for the six transfers: on RCCL they become a single grouped launch in which both directions of the xGMI
        # link are busy at once, and no ordering between the partners' sends and receives can block
        ops = [dist.P2POp(dist.irecv, t, p) for t in recvs] + [dist.P2POp(dist.isend, t, p) for t in sends]
        return list(dist.batch_isend_irecv(ops))
    works = []
    # plain point-to-point calls are matched in order on both sides: the left-eye rank posts its receives first, the
    # right-eye rank its sends
    if eye_of(rank) == 0:
        works += [dist.irecv(t, p) for t in recvs]
        works += [dist.isend(t, p) for t in sends]
    else:
        works += [dist.isend(t, p) for t in sends]
        works += [dist.irecv(t, p) for t in recvs]
    return works


def finish(works):
    """Host-level completion of exchange_async handles (Work.wait() alone only orders torch's current stream)."""
    import torch
    for w in works:
        w.wait()
    if works and torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()
