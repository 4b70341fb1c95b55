"""One-off fuzz of the stereo association: random hand-placed keypoints (all octaves, anywhere incl. borders), planted
matches + random distractors, tight and padded pitches, against the oracle.
usage (GPU box): python tools/fuzz_stereo.py [seed] [cases] [pairs]   (pairs > 4 takes the four-keypoints-per-wave kernel;
the hand-placed keypoints sit in the LAST pair of the batch: the end of every level plane)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch, msorb, orb_oracle
from msorb import synth
import matcher_cases as mc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cfg = synth.KITTI
mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
IL, IR = 2 * P - 2, 2 * P - 1
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    host = synth.stereo_batch(P, cfg["rows"], cfg["cols"], seed0=300 + it)
    pad = int(rng.choice([0, 3, 39]))
    pitch = cfg["cols"] + pad
    ex = msorb.ORBextractor(2000, 1.2, 8, 20, 7)
    flat = torch.zeros(2 * P * cfg["rows"] * pitch + 16, dtype=torch.uint8, device="cuda")
    view = flat[:2 * P * cfg["rows"] * pitch].view(2 * P, cfg["rows"], pitch)[:, :, :cfg["cols"]]
    view.copy_(torch.from_numpy(host).cuda())
    counts, _, d_kps, d_desc = ex.extract_batch(view)
    scale = np.asarray(ex.GetScaleFactors(), np.float32); inv = np.asarray(ex.GetInverseScaleFactors(), np.float32)
    n = 1800
    K = orb_oracle.KP_DTYPE
    kpl, kpr = np.zeros(n, K), np.zeros(n, K)
    octs = rng.integers(0, 8, n)
    for arr in (kpl, kpr):
        arr["size"] = 31; arr["angle"] = 0; arr["response"] = 50; arr["class_id"] = -1
    for i in range(n):
        o = int(octs[i]); lv = ex.debug_level(IL, o) if i < 8 else None
    dims = [ex.debug_level(IL, o).shape for o in range(8)]
    for i in range(n):
        o = int(octs[i]); rows_o, cols_o = dims[o]
        edge = rng.random() < 0.4
        sv = int(rng.integers(0, rows_o)) if not edge else int(rng.choice([0, 3, 5, 6, rows_o - 7, rows_o - 6, rows_o - 5, rows_o - 1]))
        suL = int(rng.integers(0, cols_o)) if not edge else int(rng.choice([0, 4, 5, 9, 10, 11, cols_o - 12, cols_o - 11, cols_o - 6, cols_o - 5, cols_o - 1]))
        disp = int(rng.integers(0, 40))
        suR = max(0, suL - disp)
        kpl[i]["x"], kpl[i]["y"], kpl[i]["octave"] = np.float32(suL) * scale[o], np.float32(sv) * scale[o], o
        kpr[i]["x"], kpr[i]["y"], kpr[i]["octave"] = np.float32(suR) * scale[o], np.float32(sv) * scale[o] + np.float32(rng.uniform(-1.5, 1.5)), int(np.clip(o + rng.integers(-1, 2), 0, 7))
    desc_l = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    desc_r = desc_l.copy()
    flip = rng.integers(0, 90, n)
    for i in range(n):
        bits = rng.choice(256, int(flip[i]), replace=False)
        for b in bits: desc_r[i, b >> 3] ^= np.uint8(1 << (b & 7))
    kps_np = d_kps.cpu().numpy().copy(); desc_np = d_desc.cpu().numpy().copy()
    kps_np[IL, :n] = kpl.view(np.uint8).reshape(n, 28); desc_np[IL, :n] = desc_l
    kps_np[IR, :n] = kpr.view(np.uint8).reshape(n, 28); desc_np[IR, :n] = desc_r
    cnt = counts.copy(); cnt[IL] = cnt[IR] = n
    d_ur, d_dp, oob, _ = msorb.stereo_matches_batch(ex, cnt, torch.from_numpy(kps_np).cuda(), torch.from_numpy(desc_np).cuda(), mb, mbf)
    pl = [ex.debug_level(IL, l) for l in range(8)]; pr = [ex.debug_level(IR, l) for l in range(8)]
    rur, rdp, roob = orb_oracle.compute_stereo_matches(kpl, desc_l, kpr, desc_r, pl, pr, scale, inv, mb, mbf)
    ur, dp = d_ur.cpu().numpy()[P - 1, :n], d_dp.cpu().numpy()[P - 1, :n]
    ok = np.array_equal(ur.view(np.uint32), rur.view(np.uint32)) and np.array_equal(dp.view(np.uint32), rdp.view(np.uint32)) and oob[P - 1] == roob
    if not ok:
        for i in np.flatnonzero(ur.view(np.uint32) != rur.view(np.uint32)):
            o = int(kpl[i]["octave"]); print("  diff kp", i, "oct", o, "dims", dims[o], "L", kpl[i]["x"] / scale[o], kpl[i]["y"] / scale[o], "R", kpr[i]["x"] / scale[o], kpr[i]["y"]/scale[o], "got", ur[i], "want", rur[i])
    print(it, "pitch", pitch, "matched", int((rur > 0).sum()), "oob", int(roob), "OK" if ok else "MISMATCH", flush=True)
    bad += 0 if ok else 1
    ex.close()
print("bad", bad)
