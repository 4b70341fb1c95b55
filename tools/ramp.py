import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd")]
import msorb
from msorb import synth
cfg = synth.KITTI
B = 128
base = synth.stereo_batch(8, cfg["rows"], cfg["cols"], seed0=0)
host = np.concatenate([base] * (B // 8 + 1))[:2 * B]
pitch = (cfg["cols"] + 63) & ~63
st = torch.zeros((2 * B, cfg["rows"], pitch), dtype=torch.uint8, device="cuda")
img = st[:, :, :cfg["cols"]]
img.copy_(torch.from_numpy(host).cuda())
exs = [msorb.ORBextractor(2000, 1.2, 8, 20, 7) for _ in range(2)]
for e in exs: e.set_overlap(1, True)
cap = exs[0].capacity
outs = [(torch.empty((2 * B, cap, 28), dtype=torch.uint8, device="cuda"), torch.empty((2 * B, cap, 32), dtype=torch.uint8, device="cuda")) for _ in range(2)]
torch.cuda.synchronize()
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
time.sleep(idle)
if len(sys.argv) > 3:
    a = torch.randn(4096, 4096, device="cuda"); te = time.perf_counter() + float(sys.argv[3])
    while time.perf_counter() < te:
        a = (a @ a).clamp_(-1, 1)
    torch.cuda.synchronize()
infl = []
stag = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t = []
t0 = time.perf_counter()
for k in range(80):
    if len(infl) == 2:
        infl.pop(0).extract_batch_wait()
        t.append(time.perf_counter())
    if k == 1 and stag > 0:
        te = time.perf_counter() + stag
        while time.perf_counter() < te: pass
    exs[k & 1].extract_batch_submit(img, (0, 0), out=outs[k & 1])
    infl.append(exs[k & 1])
while infl:
    infl.pop(0).extract_batch_wait(); t.append(time.perf_counter())
d = np.diff(np.array([t0] + t)) * 1e3
print("idle %.1fs stagger %.2f ms: per-step ms:" % (idle, stag*1e3), " ".join("%.2f" % x for x in d[:24]), "...", " ".join("%.2f" % x for x in d[60:72]), "mean last 30: %.3f" % d[-31:-1].mean())
