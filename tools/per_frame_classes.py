"""per-frame latency on the three input classes of msorb/synth.py (ms one image, ms two images, ms stereo frame)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ms-slam_amd")]
import msorb
from msorb import synth
from bench_legs.per_frame import per_frame_leg
cfg = synth.KITTI
ex = msorb.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
for tex in ("low", "default", "high", "low"):
    L, R = synth.stereo_pair(1000, cfg["rows"], cfg["cols"], texture=tex)
    r = per_frame_leg(msorb, ex, L, R)
    r = per_frame_leg(msorb, ex, L, R)
    print(tex, r["ms_one_image"], r["ms_two_images_one_call_no_match"], r["ms_stereo_frame_one_call"], r["keypoints_stereo_frame"], flush=True)
