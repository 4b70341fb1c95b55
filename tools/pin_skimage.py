#!/usr/bin/env python3
"""Independent cross-check of the oracle's OpenCV restatements against scikit-image / scipy — third-party
implementations that ARE present in the build image (/opt/conda/bin/python3.9: scikit-image 0.18.3, scipy 1.7.1), unlike
OpenCV itself.  They are not OpenCV, so they cannot pin every bit (tools/pin_opencv.py is the kit for that), but they pin
what the restatement could most plausibly get wrong:

  fast9   skimage.feature.corner_fast(img, n=9, threshold=t): the FAST-9/16 segment test of the original Rosten code with strict
          comparisons — the DETECTION SET at t must be identical.  Run at a ladder of thresholds it also pins OpenCV's score
          definition: cornerScore = the largest t at which the pixel is still a corner, i.e. mask(t) == (score >= t) for all t.
  resize  skimage.transform.resize(order=1, anti_aliasing=False): float bilinear with half-pixel centres — pins the sampling
          geometry ((x + 0.5) * scale - 0.5, edge clamp) of SURVEY Appendix A.3 to within the fixed-point rounding (+-1 level).
  blur    scipy.ndimage.gaussian_filter(sigma=2, truncate=1.5, mode='mirror'): float 7-tap Gaussian with reflect-101 borders —
          pins kernel shape and border rule of A.5 to within the 8-bit kernel quantisation (+-2 levels).
  atan2   numpy.arctan2 in degrees — pins A.6's polynomial to the 0.3 degrees OpenCV documents for fastAtan2.

Usage:  /opt/conda/bin/python3.9 tools/pin_skimage.py    -> tests/golden/skimage_pins/*.npz + meta.json
The consumer is tests/test_oracle_pins.py (CPU suite).  Inputs come from tools/pin_opencv.py's integer-hash generator."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pin_opencv as kit  # noqa: E402  (input generators only)

OUT = os.path.join(ROOT, "tests", "golden", "skimage_pins")
FAST_LADDER = (7, 8, 12, 20, 21, 33, 50, 80, 120)
RESIZE_CASES = [("kitti_l6_l7", 126, 416, 105, 346), ("small_odd", 37, 53, 31, 44), ("euroc_l6_l7", 161, 252, 134, 210)]
BLUR_CASES = [("kitti_l7", 105, 346), ("tiny", 9, 12), ("euroc_l7", 134, 210)]


def main():
    import scipy
    import scipy.ndimage as ndi
    import skimage
    from skimage.feature import corner_fast
    from skimage.transform import resize
    os.makedirs(OUT, exist_ok=True)
    meta = {"skimage": skimage.__version__, "scipy": scipy.__version__, "numpy": np.__version__,
            "python": sys.version.split()[0], "fast_ladder": list(FAST_LADDER)}

    # ---- FAST-9 detection masks over a threshold ladder ----
    rois = kit.fast_inputs()
    img = rois[kit.FAST_CELLS]                       # the 160 x 240 image
    cells = rois[:8]
    arrs = {}
    for t in FAST_LADDER:
        arrs[f"image_t{t}"] = np.packbits(corner_fast(img.astype(np.float64), 9, float(t)) > 0)
    for i, c in enumerate(cells):
        for t in (7, 20):
            arrs[f"cell{i}_t{t}"] = np.packbits(corner_fast(c.astype(np.float64), 9, float(t)) > 0)
    np.savez_compressed(os.path.join(OUT, "fast9_masks.npz"), **arrs)
    meta["fast_inputs_sha256"] = kit.sha(np.concatenate([img.ravel()] + [c.ravel() for c in cells]))

    # ---- float bilinear resize (values * 8, uint16) ----
    arrs = {}
    for i, (name, sr, sc, dr, dc) in enumerate(RESIZE_CASES):
        src = kit.pin_image(300 + i, sr, sc)
        dst = resize(src.astype(np.float64), (dr, dc), order=1, mode="edge", anti_aliasing=False, preserve_range=True, clip=False)
        arrs[name] = np.rint(dst * 8).astype(np.uint16)
    np.savez_compressed(os.path.join(OUT, "resize_float.npz"), **arrs)

    # ---- float Gaussian sigma 2, radius 3, mirror borders (values * 8, uint16) ----
    arrs = {}
    for i, (name, r, c) in enumerate(BLUR_CASES):
        src = kit.pin_image(400 + i, r, c)
        dst = ndi.gaussian_filter(src.astype(np.float64), sigma=2.0, truncate=1.5, mode="mirror")
        arrs[name] = np.rint(dst * 8).astype(np.uint16)
    np.savez_compressed(os.path.join(OUT, "gaussian_float.npz"), **arrs)

    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", OUT, meta)


if __name__ == "__main__":
    main()
