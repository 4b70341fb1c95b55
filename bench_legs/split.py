"""configs[3]: self-validation of the stereo L/R split after a --gpus N run."""
import os
import sys
import time

import numpy as np

from . import ROOT, KITTI_MB, KITTI_MBF, self_check, oracle_module, tests_dir


def split_self_validation(msorb, torch, dist, stereo_split, rank, world, eye, half, ex_own, ex_other, make_ex, images, other_images,
                          mine, theirs, rank_value, dev, rank_keypoints_per_step=None):
    """Untimed, after the timed region of a --gpus N run: (1) who took part (ranks, devices, backend), (2) one more exchange, after
    which every rank checks the features it RECEIVED for its pairs against a local extraction of the same images (it holds the
    other eye's images of the pairs it joins) — bit for bit —, and that the stereo association on (own, gathered) features
    equals the association on (own, locally extracted) ones: the split path gives the single-GPU result; (3) per-rank rates.
    An assertion failure here aborts the run: a wrong 2-GPU number is never printed."""
    backend = dist.get_backend()
    info = [None] * world
    dist.all_gather_object(info, {"rank": rank, "device": torch.cuda.get_device_name(dev), "cuda_index": dev.index,
                                  "mkeypoints_per_s": round(rank_value / 1e6, 3), "keypoints_per_step": rank_keypoints_per_step})
    ones = torch.ones(1, dtype=torch.int32, device=dev)
    dist.all_reduce(ones)
    # fresh extraction of this rank's images + exchange
    counts, _, _, _ = ex_own.extract_batch(images, (0, 0), out=(mine.kps, mine.desc))
    mine.counts.copy_(torch.from_numpy(counts))
    works = stereo_split.swap_halves_async(dist, rank, world, mine, theirs)
    stereo_split.finish(works)
    same_features = same_assoc = None
    if works:
        ex_chk = make_ex()
        try:
            c_loc, _, k_loc, d_loc = ex_chk.extract_batch(other_images)          # the other eye of my pairs, extracted HERE
            got_c = theirs.counts.cpu().numpy()
            same_features = bool(np.array_equal(got_c, c_loc[:half]))
            for i in range(half):
                n = int(c_loc[i])
                same_features = same_features and bool(torch.equal(theirs.kps[i, :n], k_loc[i, :n]) and torch.equal(theirs.desc[i, :n], d_loc[i, :n]))
            own = (mine.counts[:half], mine.kps[:half], mine.desc[:half])
            got = (theirs.counts, theirs.kps, theirs.desc)
            loc = (torch.from_numpy(np.ascontiguousarray(c_loc[:half])).to(dev), k_loc[:half], d_loc[:half])
            ex_other.pyramid_batch(other_images)
            if eye == 0:
                ur_split = msorb.stereo_matches_split(ex_own, ex_other, *own, *got, KITTI_MB, KITTI_MBF)[0]
                ur_local = msorb.stereo_matches_split(ex_own, ex_chk, *own, *loc, KITTI_MB, KITTI_MBF)[0]
            else:
                ur_split = msorb.stereo_matches_split(ex_other, ex_own, *got, *own, KITTI_MB, KITTI_MBF)[0]
                ur_local = msorb.stereo_matches_split(ex_chk, ex_own, *loc, *own, KITTI_MB, KITTI_MBF)[0]
            same_assoc = bool(torch.equal(ur_split, ur_local))
            matched = int((ur_split > 0).sum().item())
        finally:
            ex_chk.close()
        assert same_features, f"rank {rank}: the gathered features differ from a local extraction of the same images"
        assert same_assoc, f"rank {rank}: the split stereo association differs from the single-GPU association"
    flags = [None] * world
    dist.all_gather_object(flags, {"rank": rank, "gathered_features_equal_local": same_features, "split_association_equals_local": same_assoc,
                                   "matched": matched if works else None})
    return {"backend": backend, "rccl": backend == "nccl", "ranks_seen": int(ones.item()), "ranks": info,
            "bytes_exchanged_per_step_and_rank": {"sent": sum(t[half:].numel() * t.element_size() for t in mine.tensors()),
                                                  "received": theirs.nbytes()},
            "checks": flags}
