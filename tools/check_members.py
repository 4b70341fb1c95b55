#!/usr/bin/env python3
"""Does the host layer (ms-slam_amd/host/*) only touch members the reference really has, with access it really grants?

The drop-in classes and the msorb_host:: templates are compiled here against hand-written stand-ins (tests/slam_stub,
tests/cv_stub): OpenCV / Eigen / Sophus are not installed, so the reference's own headers cannot be compiled.  This tool closes
the gap that leaves — a member-name, access or arity drift between the stand-ins and /root/reference/include would otherwise
surface only at integration ("weak" item 3):

  1. it parses the class bodies of include/{Frame,KeyFrame,MapPoint,ORBextractor,ORBmatcher,GeometricCamera}.h of the
     reference (text level: comments stripped, braces matched, access labels tracked);
  2. it collects every `object.member` / `object->member` access in ms-slam_amd/host/*.{h,cc} whose object name identifies a
     reference type (pKF*, kf* -> KeyFrame; F, F1, F2, CurrentFrame, LastFrame -> Frame; pMP*, p -> MapPoint; *Camera ->
     GeometricCamera) and requires the member to exist in that class (or a base) and to be PUBLIC;
  3. it requires the drop-in declarations (host/ORBmatcher.h, host/ORBextractor.h) to offer every public method of the
     reference classes with the same parameter count, and the same public data members the callers read;
  4. TYPES: the declared type of every member / the return type of every method the host layer consumes (EXPECTED_TYPES:
     mvKeysUn = std::vector<cv::KeyPoint>, mDescriptors = cv::Mat, GetFeatureGrids() three vectors deep, tuple<int,int>
     observation indices, GetPose() = Sophus::SE3f ...) must be what the reference declares, and where the stand-ins of
     tests/slam_stub declare the same member, their type must be the reference's too.

Documented, deliberate additions the integration makes to the reference are listed in ALLOWED_MISSING.
Runs in the build container only (the reference is not on the GPU box); `pytest -m "not gpu"` runs it through
tests/test_check_members.py.  Exit code 0 = consistent; prints a JSON report.
"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MSORB_REFERENCE", "/root/reference")
HOST = os.path.join(ROOT, "ms-slam_amd", "host")

# members the integration ADDS to the reference (one-line accessors)
ALLOWED_MISSING = {("MapPoint", "GetMaxDistance"), ("MapPoint", "GetMinDistance")}
# object-name patterns -> reference class
# (objects held through shared_ptr / raw pointers count with `->` only: `pKF.get()`, `vpKFs.size()` are not member accesses;
# Frame objects are references: `.` only)
OBJECT_CLASS = [
    (re.compile(r"^(pKF\w*|kf|pKFi|kfs)$"), "KeyFrame", "->"),
    (re.compile(r"^(F|F1|F2|CurrentFrame|LastFrame)$"), "Frame", "."),
    (re.compile(r"^(pMP\w*|p|pMPinKF|points)$"), "MapPoint", "->"),
    (re.compile(r"^(mpCamera\w*|pCamera\w*)$"), "GeometricCamera", "->"),
]
# names that look like member accesses on those objects but are not (local structs of the host layer)
IGNORE_OBJECT_FILES = {}


def split_top_level(text):
    """split at commas that are not inside <...> or (...)"""
    out, depth, cur = [], 0, ""
    for ch in text:
        depth += {"<": 1, ">": -1, "(": 1, ")": -1}.get(ch, 0)
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return out


TYPE_ALIASES = [(r"\bSophus::SE3<float>", "Sophus::SE3f"), (r"\bSophus::Sim3<float>", "Sophus::Sim3f"), (r"\bunsignedlong(int)?\b", "longunsignedint"),
                (r"\bEigen::Matrix<float,3,1>", "Eigen::Vector3f"), (r"\bEigen::Matrix<float,3,3>", "Eigen::Matrix3f")]


def norm_type(t):
    """a declared type reduced to what a caller depends on: no storage / cv qualifiers, no references, no namespaces std / ORB_SLAM3"""
    t = re.sub(r"\b(static|inline|virtual|const|constexpr|mutable|explicit|typename|EIGEN_\w+)\b", " ", t)
    t = t.replace("&", " ")
    t = re.sub(r"\b(std|ORB_SLAM3)::", "", t)
    t = re.sub(r"\s+", "", t)
    for pat, rep in TYPE_ALIASES:
        t = re.sub(pat, rep, t)
    return t


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#[^\n]*(\\\n[^\n]*)*", " ", src, flags=re.M)
    return src


def class_bodies(src):
    """-> {name: (kind, bases, body)} for every class / struct defined at any depth of `src`."""
    out = {}
    for m in re.finditer(r"\b(class|struct)\s+(\w+)\s*(?:final\s*)?(:[^{;]*)?\{", src):
        kind, name, bases = m.group(1), m.group(2), m.group(3) or ""
        i, depth = m.end(), 1
        while i < len(src) and depth:
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        out.setdefault(name, (kind, re.findall(r"(?:public|protected|private)?\s*(\w+)\s*(?:,|$)", bases.lstrip(":")), src[m.end():i - 1]))
    return out


def members(kind, body):
    """-> {name: {"access": set, "kind": "function"/"data", "arity": set}} of one class body."""
    mem = {}
    access = "private" if kind == "class" else "public"
    i, n, stmt = 0, len(body), ""

    def flush(text, had_body=False):
        nonlocal access
        text = text.strip()
        while True:
            m = re.match(r"^(public|protected|private)\s*:(?!:)", text)
            if not m:
                break
            access = m.group(1)
            text = text[m.end():].strip()
        if not text or text.startswith(("friend", "using", "typedef", "enum", "template")) and "(" not in text:
            return
        if re.match(r"^(class|struct|enum|union)\b[^()]*$", text):
            return
        text = re.sub(r"^(template\s*<[^>]*>\s*)", "", text)
        if "(" in text and not re.match(r"^[^=(]*=", text):
            head = text[:text.index("(")]
            if "operator" in head:
                name = "operator" + head.split("operator")[1].strip()
            else:
                ids = re.findall(r"[~\w]+", head)
                if not ids:
                    return
                name = ids[-1]
            # parameter list = up to the matching ')'
            j, depth, k = text.index("("), 0, 0
            for k in range(j, len(text)):
                depth += {"(": 1, ")": -1}.get(text[k], 0)
                if depth == 0:
                    break
            params = text[j + 1:k].strip()
            if params in ("", "void"):
                arity = 0
            else:
                d2, arity = 0, 1
                for ch in params:
                    d2 += {"<": 1, ">": -1, "(": 1, ")": -1}.get(ch, 0)
                    if ch == "," and d2 == 0:
                        arity += 1
            e = mem.setdefault(name, {"access": set(), "kind": "function", "arity": set(), "types": set()})
            e["access"].add(access)
            e["arity"].add(arity)
            if "operator" not in head:   # return type = what stands in front of the name
                rt = head[:head.rindex(name)] if name in head else ""
                e["types"].add(norm_type(rt))
            return
        # data members: "type a, *b = 0, c[3];"
        text = re.sub(r"=[^,;]*", "", text)
        text = re.sub(r"\[[^\]]*\]", "", text)
        full = text
        text = re.sub(r"<[^<>]*>", " ", text)
        text = re.sub(r"<[^<>]*>", " ", text)
        text = re.sub(r"<[^<>]*>", " ", text)
        parts = text.split(",")
        # the declared type: everything of the first declarator but its name (template arguments kept)
        first_ids = re.findall(r"\w+", parts[0])
        base_type = ""
        if len(first_ids) >= 2:
            head0 = split_top_level(full)[0]
            base_type = head0[:head0.rindex(first_ids[-1])]
        for k, part in enumerate(parts):
            ids = re.findall(r"\w+", part)
            if not ids:
                continue
            name = ids[-1]
            if k == 0 and len(ids) < 2:
                continue   # a lone identifier is a macro (EIGEN_MAKE_ALIGNED_OPERATOR_NEW)
            e = mem.setdefault(name, {"access": set(), "kind": "data", "arity": set(), "types": set()})
            e["access"].add(access)
            ptr = "*" if "*" in part else ""
            bt = re.sub(r"[*&\s]+$", "", base_type) if k else base_type
            e["types"].add(norm_type(bt + (ptr if k else "")))

    while i < n:
        ch = body[i]
        if ch == "{":   # function body / initialiser / nested type: skip to the matching brace
            depth, j = 1, i + 1
            while j < n and depth:
                depth += {"{": 1, "}": -1}.get(body[j], 0)
                j += 1
            nested_type = re.match(r"^\s*(public\s*:|protected\s*:|private\s*:)?\s*(class|struct|enum|union)\b", stmt) and "(" not in stmt
            if nested_type:
                # skip "struct X {...} name;" up to the ';'
                while j < n and body[j] != ";":
                    j += 1
                flush(re.sub(r"\b(class|struct|enum|union)\b.*", "", stmt))
                stmt = ""
                i = j + 1
                continue
            flush(stmt, True)
            stmt = ""
            i = j
            if i < n and body[i] == ";":
                i += 1
            continue
        if ch == ";":
            flush(stmt)
            stmt = ""
        elif ch == ":" and re.search(r"\b(public|protected|private)\s*$", stmt) and (i + 1 >= n or body[i + 1] != ":"):
            lab = re.search(r"\b(public|protected|private)\s*$", stmt)
            flush(stmt[:lab.start()])          # whatever stood in front of the label (a macro without a semicolon)
            flush(lab.group(1) + ":")
            stmt = ""
        else:
            stmt += ch
        i += 1
    return mem


def parse_reference():
    classes = {}
    for h in ("Frame.h", "KeyFrame.h", "MapPoint.h", "ORBextractor.h", "ORBmatcher.h", "GeometricCamera.h"):
        path = os.path.join(REF, "include", h)
        if not os.path.exists(path):
            path = os.path.join(REF, "include", "CameraModels", h)
        src = strip_comments(open(path, errors="replace").read())
        for name, (kind, bases, body) in class_bodies(src).items():
            classes.setdefault(name, {"bases": bases, "members": members(kind, body), "file": h})
    return classes


def lookup(classes, cls, name):
    seen = set()
    stack = [cls]
    while stack:
        c = stack.pop()
        if c in seen or c not in classes:
            continue
        seen.add(c)
        if name in classes[c]["members"]:
            return classes[c]["members"][name]
        stack.extend(classes[c]["bases"])
    return None


def host_accesses():
    acc = {}
    files = sorted(glob.glob(os.path.join(HOST, "*.h")) + glob.glob(os.path.join(HOST, "*.cc")))
    for path in files:
        src = strip_comments(open(path).read())
        for m in re.finditer(r"\b(\w+)\s*(?:\[[^\]]*\])?\s*(->|\.)\s*(\w+)\s*(\()?", src):
            obj, op, member, call = m.group(1), m.group(2), m.group(3), bool(m.group(4))
            for pat, cls, want_op in OBJECT_CLASS:
                if pat.match(obj) and op == want_op:
                    if cls == "MapPoint" and obj == "p" and not re.match(r"^(m[A-Z_a-z]|Get|is|Is|Observations|IncreaseVisible|PredictScale|Replace|AddObservation)", member):
                        break   # `p` is also used for plain structs in the host layer
                    acc.setdefault((cls, member), set()).add((os.path.basename(path), call))
                    break
    return acc


def dropin_check(classes):
    """host/ORBmatcher.h and host/ORBextractor.h against the reference declarations."""
    problems = []
    for cls, hdr in (("ORBmatcher", "ORBmatcher.h"), ("ORBextractor", "ORBextractor.h")):
        src = strip_comments(open(os.path.join(HOST, hdr)).read())
        mine = {n: members(k, b) for n, (k, _, b) in class_bodies(src).items()}.get(cls)
        if mine is None:
            problems.append(f"{hdr}: class {cls} not declared")
            continue
        for name, e in classes[cls]["members"].items():
            if "public" not in e["access"] or name.startswith("~"):
                continue
            if name not in mine:
                problems.append(f"{hdr}: public member {cls}::{name} of the reference is missing")
                continue
            if "public" not in mine[name]["access"]:
                problems.append(f"{hdr}: {cls}::{name} is not public here")
            if e["kind"] == "function" and not e["arity"] <= mine[name]["arity"]:
                problems.append(f"{hdr}: {cls}::{name} takes {sorted(e['arity'])} parameters in the reference, {sorted(mine[name]['arity'])} here")
    return problems


# Types the host layer's code depends on (it copies vectors of 28-byte cv::KeyPoint, reads 32-byte cv::Mat rows, walks the
# feature grid three levels deep, unpacks tuple<int,int> observation indices, takes Tcw as a Sophus::SE3f ...): the declared
# type of a member / the return type of a method in the REFERENCE, normalised by norm_type().  Members the stand-ins of
# tests/slam_stub also declare are compared with those as well, so a drift on either side fails here.
EXPECTED_TYPES = {
    ("Frame", "N"): "int", ("Frame", "Nleft"): "int", ("Frame", "mnId"): "longunsignedint",
    ("Frame", "mvKeys"): "vector<cv::KeyPoint>", ("Frame", "mvKeysUn"): "vector<cv::KeyPoint>", ("Frame", "mvKeysRight"): "vector<cv::KeyPoint>",
    ("Frame", "mDescriptors"): "cv::Mat", ("Frame", "mDescriptorsRight"): "cv::Mat",
    ("Frame", "mvuRight"): "vector<float>", ("Frame", "mvDepth"): "vector<float>",
    ("Frame", "mvpMapPoints"): "vector<shared_ptr<MapPoint>>", ("Frame", "mvbOutlier"): "vector<bool>",
    ("Frame", "mvScaleFactors"): "vector<float>", ("Frame", "mb"): "float", ("Frame", "mbf"): "float",
    ("Frame", "mnMinX"): "float", ("Frame", "mnMaxX"): "float", ("Frame", "mnMinY"): "float", ("Frame", "mnMaxY"): "float",
    ("Frame", "mpCamera"): "GeometricCamera*", ("Frame", "GetPose"): "Sophus::SE3f",
    ("Frame", "mFeatVec"): "DBoW2::FeatureVector", ("Frame", "mBowVec"): "DBoW2::BowVector",
    ("Frame", "mmProjectPoints"): "map<longunsignedint,cv::Point2f>",
    ("KeyFrame", "GetAllKeyUn"): "vector<cv::KeyPoint>", ("KeyFrame", "GetDescriptor"): "cv::Mat", ("KeyFrame", "GetuRight"): "float",
    ("KeyFrame", "GetMapPoint"): "shared_ptr<MapPoint>", ("KeyFrame", "GetMapPointMatches"): "vector<shared_ptr<MapPoint>>",
    ("KeyFrame", "GetFeatureVector"): "DBoW2::FeatureVector", ("KeyFrame", "GetFeatureGrids"): "vector<vector<vector<size_t>>>",
    ("KeyFrame", "GetNumberMPs"): "int", ("KeyFrame", "GetPose"): "Sophus::SE3f", ("KeyFrame", "GetNLeft"): "int",
    ("KeyFrame", "mnId"): "longunsignedint", ("KeyFrame", "mnMinX"): "int", ("KeyFrame", "mnMaxX"): "int",
    ("KeyFrame", "mvScaleFactors"): "vector<float>", ("KeyFrame", "mbSparsified"): "bool",
    ("MapPoint", "GetWorldPos"): "Eigen::Vector3f", ("MapPoint", "GetNormal"): "Eigen::Vector3f", ("MapPoint", "GetDescriptor"): "cv::Mat",
    ("MapPoint", "Observations"): "int", ("MapPoint", "isBad"): "bool", ("MapPoint", "GetIndexInKeyFrame"): "tuple<int,int>",
    ("MapPoint", "GetObservations"): "map<shared_ptr<KeyFrame>,tuple<int,int>>",
    ("MapPoint", "mTrackProjX"): "float", ("MapPoint", "mTrackProjXR"): "float", ("MapPoint", "mbTrackInView"): "bool",
    ("MapPoint", "mnTrackScaleLevel"): "int", ("MapPoint", "mnLastFrameSeen"): "longunsignedint", ("MapPoint", "mbSparsified"): "bool",
    ("GeometricCamera", "project"): "Eigen::Vector2f",
}


def parse_stub():
    path = os.path.join(ROOT, "tests", "slam_stub", "slam_stub_types.h")
    src = strip_comments(open(path).read())
    return {n: {"bases": b, "members": members(k, body), "file": "slam_stub_types.h"} for n, (k, b, body) in class_bodies(src).items()}


def type_check(classes, accesses):
    """-> (problems, number of type comparisons made)"""
    problems, n = [], 0
    stub = parse_stub()
    for key, want in sorted(EXPECTED_TYPES.items()):
        e = lookup(classes, *key)
        n += 1
        if e is None:
            problems.append(f"{key[0]}::{key[1]}: the host layer expects type {want}, the reference has no such member")
        elif want not in e["types"]:
            problems.append(f"{key[0]}::{key[1]}: the host layer expects type {want}, the reference declares {sorted(e['types'])}")
    for (cls, member) in sorted(accesses):
        r, t = lookup(classes, cls, member), lookup(stub, cls, member)
        if r is None or t is None:
            continue
        n += 1
        if not t["types"] <= r["types"]:
            problems.append(f"{cls}::{member}: the stand-in (tests/slam_stub) declares {sorted(t['types'])}, the reference {sorted(r['types'])}")
    return problems, n


def main():
    if not os.path.isdir(os.path.join(REF, "include")):
        print(json.dumps({"skipped": f"{REF}/include not present (the reference exists in the build container only)"}))
        return 0
    classes = parse_reference()
    problems, checked = [], 0
    for (cls, member), uses in sorted(host_accesses().items()):
        if (cls, member) in ALLOWED_MISSING:
            continue
        e = lookup(classes, cls, member)
        checked += 1
        where = ", ".join(sorted({u[0] for u in uses}))
        if e is None:
            problems.append(f"{cls}::{member} (used in {where}) does not exist in the reference")
        elif "public" not in e["access"]:
            problems.append(f"{cls}::{member} (used in {where}) is {'/'.join(sorted(e['access']))} in the reference")
        elif any(u[1] for u in uses) and e["kind"] != "function":
            problems.append(f"{cls}::{member} is called like a function in {where} but is a data member in the reference")
    problems += dropin_check(classes)
    type_problems, n_types = type_check(classes, host_accesses())
    problems += type_problems
    report = {"type_comparisons": n_types, "reference": REF, "classes_parsed": {c: len(v["members"]) for c, v in classes.items() if c in
                                                  ("Frame", "KeyFrame", "MapPoint", "ORBextractor", "ORBmatcher", "GeometricCamera")},
              "member_accesses_checked": checked, "allowed_additions": sorted("::".join(a) for a in ALLOWED_MISSING), "problems": problems}
    print(json.dumps(report, indent=1))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
