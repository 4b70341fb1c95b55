#!/usr/bin/env python3
"""Generate tests/golden/extractor_golden.json (+ .npz heads) from the CPU oracle.

The reference itself cannot run here (no OpenCV), so these are ORACLE goldens: they freeze the oracle's
output for seeded synthetic images, recording the semantics version (Gaussian kernel etc.)."""
import hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "oracle")]
from msorb import synth
import orb_oracle

CASES = [dict(seed=1000, rows=240, cols=320, nfeatures=500, lapping=[0, 0]),
         dict(seed=1001, rows=480, cols=752, nfeatures=1000, lapping=[0, 0]),
         dict(seed=1002, rows=376, cols=1241, nfeatures=2000, lapping=[0, 0]),
         dict(seed=1003, rows=400, cols=800, nfeatures=2000, lapping=[0, 1000])]

def main():
    out = dict(semantics="SURVEY.md Appendix A; fma(x,b,y*a)/fma(x,a,-(y*b)); glibc sinf/cosf",
               gauss_kernel_q88=[18, 34, 48, 56, 48, 34, 18], cases=[])
    gd = os.path.join(ROOT, "tests", "golden")
    for c in CASES:
        img = synth.image(c["seed"], c["rows"], c["cols"])
        ex = orb_oracle.OracleExtractor(c["nfeatures"], 1.2, 8, 20, 7)
        mono, kps, desc = ex(img, tuple(c["lapping"]))
        h = hashlib.sha256()
        h.update(np.int32(mono).tobytes()); h.update(kps.view(np.uint8).tobytes()); h.update(desc.tobytes())
        head = f"head_{c['seed']}.npz"
        np.savez_compressed(os.path.join(gd, head), kps=kps[:16].view(np.uint8).reshape(16, 28), desc=desc[:16])
        out["cases"].append(dict(c, image_sha256=hashlib.sha256(img.tobytes()).hexdigest(), n_keypoints=int(len(kps)),
                                 mono_index=int(mono), digest=h.hexdigest(), head_file=head))
    json.dump(out, open(os.path.join(gd, "extractor_golden.json"), "w"), indent=1)

if __name__ == "__main__":
    main()
