"""Frame::ComputeStereoMatches for the whole batch, device resident (pair p = images 2p / 2p+1), median rejection included;
the outputs of the extraction are its inputs."""
import numpy as np

from . import KITTI_MB, KITTI_MBF


def stereo_leg(msorb, ex, counts_h, d_kps, d_desc):
    """-> (the `stereo_match` object, d_ur of the batch)."""
    msorb.stereo_matches_batch(ex, counts_h, d_kps, d_desc, KITTI_MB, KITTI_MBF)
    sms = [msorb.stereo_matches_batch(ex, counts_h, d_kps, d_desc, KITTI_MB, KITTI_MBF)[3] for _ in range(15)]
    d_ur, _, _, _ = msorb.stereo_matches_batch(ex, counts_h, d_kps, d_desc, KITTI_MB, KITTI_MBF)
    m = float(np.median(sms))
    n_left = int(counts_h[0::2].sum())
    stereo = {"pairs": int(len(counts_h) // 2), "left_keypoints": n_left, "matched": int((d_ur > 0).sum().item()),
              "ms_per_batch": round(m, 4), "mkeypoints_per_s": round(n_left / (m * 1e-3) / 1e6, 2),
              "kernels": "stereo_rowtable_kernel + stereo_match_quad_kernel (four left keypoints per wave) + stereo_median_kernel"}
    return stereo, d_ur
