"""The extractor's constants held to values recorded INDEPENDENTLY of the tables the product and the oracle compile in.

Oracle and kernels share ms-slam_amd/csrc/orb_pattern.inc and take their scale / quota / umax / level-size tables from this
repository's code: a wrong shared constant would pass every GPU-vs-oracle comparison.  tests/golden/reference_constants.json is
derived a second way (tools/make_reference_constants.py: the pattern from the reference's TEXT, the rest from the constructor's
formulas restated in numpy).  Here:
  CPU  the fixture == the reference text / a fresh run of the tool (when /root/reference is mounted), == SURVEY.md section 8's table
       (the surveyor's own computation: umax, quotas, level sizes), == the three copies of the pattern the repo carries
       (csrc/orb_pattern.inc, the pin kit's embedded copy) and the oracle's constructor tables;
  GPU  the fixture == what the DEVICE holds: the __constant__ pattern and umax read back from the chip
       (msorb_debug_patch_tables), msorb_extractor_tables' float bit patterns, features per level, level sizes of a run."""
import hashlib
import importlib.util
import json
import os
import re

import numpy as np
import pytest

from msorb import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constants.json")))
# SURVEY.md section 8 (config table and row a1), typed in from the survey — not computed by anything in this repository
SURVEY_UMAX = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
SURVEY = {"kitti": dict(sizes=[(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)],
                        quota=[434, 362, 302, 251, 209, 175, 145, 122]),
          "euroc": dict(sizes=[(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)],
                        quota=[261, 217, 181, 151, 126, 105, 87, 72]),
          "euroc_1000": dict(sizes=None, quota=[217, 181, 151, 126, 105, 87, 73, 60]),
          "4seasons": dict(sizes=[(800, 400), (667, 333), (556, 278), (463, 231), (386, 193), (322, 161), (268, 134), (223, 112)],
                           quota=[434, 362, 302, 251, 209, 175, 145, 122])}
SURVEY_SCALE = [1.0, 1.2000000477, 1.4400000572, 1.7280001640, 2.0736002922, 2.4883203506, 2.9859845638, 3.5831816196]


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _inc_pattern():
    txt = re.sub(r"//.*", "", open(os.path.join(ROOT, "ms-slam_amd", "csrc", "orb_pattern.inc")).read())
    return np.array([int(x) for x in re.findall(r"-?\d+", txt)], np.int8)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, np.int8).tobytes()).hexdigest()


def test_fixture_is_the_reference_text_and_the_surveys_table():
    assert FIX["umax"] == SURVEY_UMAX
    for name, s in SURVEY.items():
        c = FIX["configs"][name]
        assert c["features_per_level"] == s["quota"], name
        assert sum(c["features_per_level"]) == c["nfeatures"]
        if s["sizes"]:
            assert [tuple(x) for x in c["level_sizes_wh"]] == s["sizes"], name
        assert np.array(c["scale_bits"], np.uint32).view(np.float32).tolist() == [float(np.float32(v)) for v in SURVEY_SCALE]
    if os.path.exists("/root/reference/src/ORBextractor.cc"):     # this container; the GPU box has the fixture only
        tool = _load(os.path.join(ROOT, "tools", "make_reference_constants.py"), "mrc")
        pat = tool.reference_pattern()
        assert _sha(pat) == FIX["pattern_sha256"] and pat[:4].tolist() == FIX["pattern_first_pair"] and pat[-4:].tolist() == FIX["pattern_last_pair"]
        assert tool.umax_table() == FIX["umax"]


def test_every_copy_of_the_pattern_in_the_repo_is_the_fixtures():
    inc = _inc_pattern()
    assert inc.shape == (1024,) and _sha(inc) == FIX["pattern_sha256"]
    assert int(inc.astype(np.int64).sum()) == FIX["pattern_sum"] and int(np.abs(inc.astype(np.int64)).sum()) == FIX["pattern_abs_sum"]
    kit = _load(os.path.join(ROOT, "tools", "pin_opencv.py"), "pin_opencv_consts")
    assert _sha(kit.orb_pattern().ravel()) == FIX["pattern_sha256"]


def test_oracle_constructor_tables_are_the_fixtures(oracle):
    for name, c in FIX["configs"].items():
        ex = oracle.OracleExtractor(c["nfeatures"], c["scale_factor"], c["nlevels"], 20, 7)
        ex(synth.image(1, c["rows"], c["cols"]))
        assert [ex.level(l).shape[::-1] for l in range(c["nlevels"])] == [tuple(x) for x in c["level_sizes_wh"]], name


def _assert_bits(got, want_bits, what):
    assert np.asarray(got, np.float32).view(np.uint32).tolist() == want_bits, what


@pytest.mark.gpu
def test_device_tables_are_the_fixtures(msorb_mod):
    """What the chip holds, not what the sources say: a table edit that changes oracle and kernel together stops here."""
    for name, c in FIX["configs"].items():
        ex = msorb_mod.ORBextractor(c["nfeatures"], c["scale_factor"], c["nlevels"], 20, 7)
        try:
            pat, umax = ex.debug_patch_tables()
            assert _sha(pat.ravel()) == FIX["pattern_sha256"], "the rBRIEF pattern in the device's constant memory is not the reference's"
            assert pat[0].tolist() == FIX["pattern_first_pair"] and pat[-1].tolist() == FIX["pattern_last_pair"]
            assert umax.tolist() == FIX["umax"]
            _assert_bits(ex.GetScaleFactors(), c["scale_bits"], "mvScaleFactor")
            _assert_bits(ex.GetInverseScaleFactors(), c["inv_scale_bits"], "mvInvScaleFactor")
            _assert_bits(ex.GetScaleSigmaSquares(), c["sigma2_bits"], "mvLevelSigma2")
            _assert_bits(ex.GetInverseScaleSigmaSquares(), c["inv_sigma2_bits"], "mvInvLevelSigma2")
            assert list(ex.features_per_level()) == c["features_per_level"], name
            ex(synth.image(1, c["rows"], c["cols"]))
            assert [ex.debug_level_size(l)[::-1] for l in range(c["nlevels"])] == [tuple(x) for x in c["level_sizes_wh"]], name
            for l in (0, 3, 7):      # and the planes the kernels produced have those sizes
                assert ex.debug_level(0, l).shape[::-1] == tuple(c["level_sizes_wh"][l])
        finally:
            ex.close()
