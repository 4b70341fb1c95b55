"""tools/check_members.py in the CPU suite: the host layer (drop-in classes + msorb_host:: templates) may only touch members that
exist and are public in /root/reference/include — it is compiled here against stand-ins, so this is the only place a drift
between the stand-ins and the real headers shows before integration.  Skips where the reference is absent (GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = os.environ.get("MSORB_REFERENCE", "/root/reference")
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include")), reason="reference headers not present")


@needs_reference
def test_host_layer_matches_reference_headers():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_members.py")], capture_output=True, text=True)
    rep = json.loads(r.stdout)
    assert rep["problems"] == [], "\n".join(rep["problems"])
    assert r.returncode == 0
    assert rep["member_accesses_checked"] >= 80
    assert rep["type_comparisons"] >= 120
    for c in ("Frame", "KeyFrame", "MapPoint", "ORBextractor", "ORBmatcher", "GeometricCamera"):
        assert rep["classes_parsed"][c] > 10


@needs_reference
def test_parser_sees_access_and_kind():
    """The checker has teeth: it tells protected from public and data from functions (KeyFrame::NLeft is protected and only
    reachable through GetNLeft(); Frame's feature arrays are public, KeyFrame's are not)."""
    import check_members as cm
    classes = cm.parse_reference()
    assert "public" not in cm.lookup(classes, "KeyFrame", "NLeft")["access"]
    e = cm.lookup(classes, "KeyFrame", "GetNLeft")
    assert e["kind"] == "function" and "public" in e["access"] and e["arity"] == {0}
    assert "public" in cm.lookup(classes, "Frame", "mvKeysUn")["access"] and cm.lookup(classes, "Frame", "mvKeysUn")["kind"] == "data"
    assert "public" not in cm.lookup(classes, "KeyFrame", "mvKeysUn")["access"]
    assert "public" not in cm.lookup(classes, "MapPoint", "mfMaxDistance")["access"]
    assert cm.lookup(classes, "MapPoint", "NoSuchMember") is None
    assert cm.lookup(classes, "ORBmatcher", "SearchByProjection")["arity"] >= {4, 5, 6, 8}
    assert "protected" in cm.lookup(classes, "ORBmatcher", "mfNNratio")["access"]
    assert cm.lookup(classes, "ORBextractor", "mvImagePyramid")["access"] == {"public"}


@needs_reference
def test_type_drift_is_caught(tmp_path, monkeypatch):
    """The type check has teeth: a copy of the reference headers in which Frame::mvKeysUn became a vector of Point2f, the
    feature grid lost a nesting level and GetIndexInKeyFrame returns a plain int must fail — member by member."""
    import shutil
    import check_members as cm
    ref2 = tmp_path / "ref"
    shutil.copytree(os.path.join(REF, "include"), ref2 / "include")
    def edit(name, old, new):
        p = ref2 / "include" / name
        src = p.read_text(errors="replace")
        assert old in src, (name, old)
        p.write_text(src.replace(old, new, 1))
    edit("Frame.h", "std::vector<cv::KeyPoint> mvKeysUn;", "std::vector<cv::Point2f> mvKeysUn;")
    edit("KeyFrame.h", "std::vector< std::vector <std::vector<size_t> > > GetFeatureGrids()", "std::vector<std::vector<size_t> > GetFeatureGrids()")
    edit("MapPoint.h", "tuple<int,int> GetIndexInKeyFrame(", "int GetIndexInKeyFrame(")
    monkeypatch.setattr(cm, "REF", str(ref2))
    classes = cm.parse_reference()
    problems, _ = cm.type_check(classes, cm.host_accesses())
    text = "\n".join(problems)
    assert "Frame::mvKeysUn" in text and "KeyFrame::GetFeatureGrids" in text and "MapPoint::GetIndexInKeyFrame" in text
    assert len(problems) >= 4      # EXPECTED_TYPES and the stand-in comparison both object to mvKeysUn
