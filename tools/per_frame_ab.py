"""per-frame latency A/B: bench_legs.per_frame on KITTI geometry, n repetitions"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ms-slam_amd")]
import msorb
from msorb import synth
from bench_legs.per_frame import per_frame_leg
cfg = synth.KITTI
L, R = synth.stereo_pair(1000, cfg["rows"], cfg["cols"])
ex = msorb.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
for i in range(3):
    r = per_frame_leg(msorb, ex, L, R)
    print(os.environ.get("MSORB_FRAME_COPIES", "blit"), r["ms_one_image"], r["ms_two_images_one_call_no_match"], r["ms_stereo_frame_one_call"], flush=True)
