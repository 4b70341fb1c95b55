#!/usr/bin/env python3
"""Constants of the extractor, recorded INDEPENDENTLY of the tables the product and the oracle compile in.

Oracle and kernels share ms-slam_amd/csrc/orb_pattern.inc and both take their scale / quota / umax tables from code of this
repository, so a wrong shared constant would be invisible to every GPU-vs-oracle test.  This script derives the same constants
a second way — the pattern's numbers from the reference's TEXT (/root/reference/src/ORBextractor.cc:149-406), umax / scale
factors / per-level quotas / level sizes from the formulas of ORBextractor.cc:409-469 and :1170-1178 restated here in numpy
float32 / float64 — and writes tests/golden/reference_constants.json.  tests/test_reference_constants.py holds the fixture to
the reference text (CPU, when the reference is mounted), to SURVEY.md section 8's table (the surveyor's own computation) and to what
the DEVICE holds (GPU: the __constant__ pattern / umax read back from the chip, msorb_extractor_tables, level sizes).

    python tools/make_reference_constants.py        # needs /root/reference (this container); the fixture travels"""
import hashlib
import json
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/ORBextractor.cc"
HALF_PATCH_SIZE = 15      # ORBextractor.cc:72


def reference_pattern():
    txt = open(REF).read()
    beg = txt.index("{", txt.index("bit_pattern_31_[256*4]"))
    body = re.sub(r"/\*.*?\*/", "", txt[beg + 1:txt.index("};", beg)], flags=re.S)
    vals = np.array([int(v) for v in re.findall(r"-?\d+", body)], np.int8)
    assert vals.shape == (1024,)
    return vals


def cv_round(x):
    return int(np.rint(x))


def umax_table():
    """ORBextractor.cc:453-468."""
    vmax = int(np.floor(np.float32(HALF_PATCH_SIZE) * np.sqrt(np.float32(2.0)) / np.float32(2) + np.float32(1)))
    vmin = int(np.ceil(np.float32(HALF_PATCH_SIZE) * np.sqrt(np.float32(2.0)) / np.float32(2)))
    hp2 = float(HALF_PATCH_SIZE * HALF_PATCH_SIZE)
    umax = [0] * (HALF_PATCH_SIZE + 1)
    for v in range(vmax + 1):
        umax[v] = cv_round(np.sqrt(hp2 - v * v))
    v, v0 = HALF_PATCH_SIZE, 0
    while v >= vmin:
        while umax[v0] == umax[v0 + 1]:
            v0 += 1
        umax[v] = v0
        v0 += 1
        v -= 1
    return umax


def ctor_tables(nfeatures, scale_factor, nlevels):
    """ORBextractor.cc:409-445: `scaleFactor` is a double member initialised from the float argument; the vectors are float."""
    sf = np.float64(np.float32(scale_factor))
    scale = [np.float32(1.0)]
    for _ in range(1, nlevels):
        scale.append(np.float32(np.float64(scale[-1]) * sf))
    sigma2 = [np.float32(1.0)] + [np.float32(s * s) for s in scale[1:]]
    inv_scale = [np.float32(1.0) / s for s in scale]
    inv_sigma2 = [np.float32(1.0) / s for s in sigma2]
    factor = np.float32(np.float64(1.0) / sf)
    # nfeatures*(1 - factor)/(1 - (float)pow((double)factor, (double)nlevels)): int * float / float, every step in float32
    n_desired = np.float32(nfeatures) * (np.float32(1) - factor) / (np.float32(1) - np.float32(np.float64(factor) ** np.float64(nlevels)))
    per_level, total = [], 0
    for _ in range(nlevels - 1):
        per_level.append(cv_round(n_desired))
        total += per_level[-1]
        n_desired = np.float32(n_desired * factor)
    per_level.append(max(nfeatures - total, 0))
    return scale, inv_scale, sigma2, inv_sigma2, per_level


def level_sizes(rows, cols, inv_scale):
    """ORBextractor.cc:1174-1175: Size(cvRound((float)cols*scale), cvRound((float)rows*scale)) from the ORIGINAL size."""
    return [[cv_round(np.float32(cols) * s), cv_round(np.float32(rows) * s)] for s in inv_scale]


def bits(v):
    return [int(np.float32(x).view(np.uint32)) for x in v]


def main():
    pat = reference_pattern()
    out = {"source": "tools/make_reference_constants.py from /root/reference/src/ORBextractor.cc (pattern: the text's numbers; the rest: "
                     "the constructor's formulas restated in numpy)",
           "pattern_sha256": hashlib.sha256(pat.tobytes()).hexdigest(), "pattern_sum": int(pat.astype(np.int64).sum()),
           "pattern_abs_sum": int(np.abs(pat.astype(np.int64)).sum()), "pattern_first_pair": pat[:4].tolist(), "pattern_last_pair": pat[-4:].tolist(),
           "umax": umax_table(), "configs": {}}
    for name, rows, cols, nfeat in (("kitti", 376, 1241, 2000), ("euroc", 480, 752, 1200), ("euroc_1000", 480, 752, 1000), ("4seasons", 400, 800, 2000)):
        scale, inv_scale, sigma2, inv_sigma2, per_level = ctor_tables(nfeat, 1.2, 8)
        out["configs"][name] = {"rows": rows, "cols": cols, "nfeatures": nfeat, "scale_factor": 1.2, "nlevels": 8,
                                "scale_bits": bits(scale), "inv_scale_bits": bits(inv_scale), "sigma2_bits": bits(sigma2),
                                "inv_sigma2_bits": bits(inv_sigma2), "features_per_level": per_level,
                                "level_sizes_wh": level_sizes(rows, cols, inv_scale)}
    path = os.path.join(ROOT, "tests", "golden", "reference_constants.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
