"""tracking_loop.unchanged_callers — a tracking frame's front-end through the UNCHANGED call sites (north_star: "drops into
Tracking.cc / LocalMapping.cc unchanged"): two std::threads x ORB_SLAM3::ORBextractor::operator() (host pyramid on), the class
ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) on its default HOST-projected path, the class
SearchByProjection(F, vpMapPoints, th, ...) — timed inside a C++ harness (tools/unchanged_callers.cc) that links the drop-in
classes of ms-slam_amd/host against libmsorb.so, each with its CPU-oracle leg and an enforced gpu_matches_cpu."""
import os
import struct
import subprocess
import tempfile
import time

import numpy as np

from . import ROOT, KITTI_MB, KITTI_MBF, self_check, oracle_module

HARNESS_SRC = os.path.join(ROOT, "tools", "unchanged_callers.cc")
HARNESS_EXE = os.path.join(ROOT, "tools", "_unchanged_callers")


def build_harness(force=False):
    """g++ of the harness + the two host classes against the stand-in headers (also run by __graft_entry__.build())."""
    deps = [HARNESS_SRC] + [os.path.join(ROOT, "ms-slam_amd", "host", f) for f in os.listdir(os.path.join(ROOT, "ms-slam_amd", "host"))] + \
           [os.path.join(ROOT, "include", "msorb.h"), os.path.join(ROOT, "tests", "slam_stub", "slam_stub_types.h")]
    if not force and os.path.exists(HARNESS_EXE) and all(os.path.getmtime(HARNESS_EXE) >= os.path.getmtime(d) for d in deps):
        return HARNESS_EXE
    subprocess.check_call(["g++", "-O2", "-std=c++17", f"-I{ROOT}/tests/slam_stub", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", HARNESS_SRC, f"{ROOT}/ms-slam_amd/host/ORBextractor.cc", f"{ROOT}/ms-slam_amd/host/ORBmatcher.cc",
                           f"-L{ROOT}/ms-slam_amd", "-lmsorb", f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
                           "-lpthread", "-o", HARNESS_EXE])
    return HARNESS_EXE


def write_scene(path, cfg, cam, left, right, ur, depth, last, mp, scratch, th_mm, th_lp):
    """the scene file of tools/unchanged_callers.cc (layout: its header comment / the rd<> calls at the top of main)"""
    NL, M = len(last["has_point"]), len(mp["obs"])
    kp = np.zeros(NL, np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"),
                                ("class_id", "<i4")]))
    kp["octave"], kp["angle"] = last["octave"], last["angle"]
    with open(path, "wb") as f:
        f.write(struct.pack("<6i", cfg["rows"], cfg["cols"], cfg["nfeatures"], cfg["nlevels"], NL, M))
        f.write(struct.pack("<13f", cfg["scale"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], KITTI_MB, KITTI_MBF, th_mm, th_lp,
                            0.0, float(cfg["cols"]), 0.0, float(cfg["rows"])))
        for a in (left, right):
            f.write(np.ascontiguousarray(a, np.uint8).tobytes())
        f.write(struct.pack("<i", len(ur)))
        for a, dt in ((ur, np.float32), (depth, np.float32), (last["Rcw"].reshape(9), np.float32), (last["tcw"], np.float32),
                      (last["Rlw"].reshape(9), np.float32), (last["tlw"], np.float32)):
            f.write(np.ascontiguousarray(a, dt).tobytes())
        f.write(kp.tobytes())
        for a, dt in ((last["has_point"], np.uint8), (np.zeros(NL), np.uint8), (last["pos_w"], np.float32), (last["obs"], np.int32),
                      (last["desc"], np.uint8), (scratch["track_in_view"], np.uint8), (mp["bad"], np.uint8), (mp["sparsified"], np.uint8),
                      (scratch["proj_x"], np.float32), (scratch["proj_y"], np.float32), (scratch["proj_xr"], np.float32),
                      (scratch["track_depth"], np.float32), (scratch["level"], np.int32), (scratch["view_cos"], np.float32),
                      (mp["desc"], np.uint8), (mp["obs"], np.int32)):
            f.write(np.ascontiguousarray(a, dt).tobytes())


def read_result(path, NL):
    b = open(path, "rb").read()
    N, NR, n14, n13, fwd, bwd, pyr_ok, iters = struct.unpack_from("<8i", b, 0)
    t_ext, t14, t13, t_tot = struct.unpack_from("<4d", b, 32)
    pos = 64
    ids14 = np.frombuffer(b, np.int32, N, pos); pos += 4 * N
    ids13 = np.frombuffer(b, np.int32, N, pos); pos += 4 * N
    valid = np.frombuffer(b, np.uint8, NL, pos); pos += NL
    u = np.frombuffer(b, np.float32, NL, pos); pos += 4 * NL
    v = np.frombuffer(b, np.float32, NL, pos); pos += 4 * NL
    ur = np.frombuffer(b, np.float32, NL, pos); pos += 4 * NL
    kps = np.frombuffer(b, np.uint8, 28 * N, pos).reshape(N, 28); pos += 28 * N
    desc = np.frombuffer(b, np.uint8, 32 * N, pos).reshape(N, 32); pos += 32 * N
    assert pos == len(b), (pos, len(b))
    return dict(N=N, NR=NR, n14=n14, n13=n13, fwd=fwd, bwd=bwd, pyr_ok=pyr_ok, iters=iters, ms_extract=t_ext, ms_a14=t14, ms_a13=t13,
                ms_total=t_tot, ids14=ids14, ids13=ids13, valid=valid, u=u, v=v, ur=ur, kps=kps, desc=desc)


def unchanged_callers_leg(msorb, synth, cfg, left, right, kps, desc, ur, depth, scale, mp, frustum, last, th_mm, th_lp, cpu, iters=60,
                          device=0):
    """kps / desc / ur / depth: the left eye's features of (left, right) as the batch path returned them (the harness re-extracts
    through the class and must find the same ones); mp / frustum: the local map of SearchLocalPoints; last: synth.last_frame's
    table with its poses."""
    orb_oracle = oracle_module()
    exe = build_harness()
    cam = synth.KITTI_CAM
    bounds = (0.0, float(cfg["cols"]), 0.0, float(cfg["rows"]))
    NL, M, N = len(last["has_point"]), len(mp["obs"]), len(kps)
    # Frame::isInFrustum stays HOST code with unchanged callers (Tracking.cc:3343-3361): its scratch comes from the oracle
    r = orb_oracle.is_in_frustum(frustum, mp["pos_w"], mp["normal"], mp["max_distance"], mp["min_distance"], 0.5)
    scratch = dict(r)
    scratch["track_in_view"] = (r["track_in_view"].astype(bool) & mp["visit"].astype(bool)).astype(np.uint8)
    with tempfile.TemporaryDirectory(prefix="msorb_uc_") as d:
        scene, out = os.path.join(d, "scene.bin"), os.path.join(d, "out.bin")
        write_scene(scene, cfg, cam, left, right, ur, depth, last, mp, scratch, th_mm, th_lp)
        env = dict(os.environ, MSORB_DEVICE=str(device))
        env.pop("MSORB_DEVICES", None)
        env.pop("MSORB_HOST_PYRAMID", None)      # the default: mvImagePyramid is copied back for the host's ComputeStereoMatches
        p = subprocess.run([exe, scene, out, str(iters)], env=env, capture_output=True, text=True, timeout=300)
        if p.returncode != 0:
            raise RuntimeError(f"tools/_unchanged_callers exited with {p.returncode}: {p.stderr[-400:]}")
        g = read_result(out, NL)
    # the class extracted what the batch path extracted (same images, same library)
    self_check(g["N"] == N and np.array_equal(g["kps"], np.ascontiguousarray(kps).view(np.uint8).reshape(N, 28)) and
               np.array_equal(g["desc"], desc), "unchanged_callers: ORBextractor::operator() and msorb_extract_batch disagree")
    self_check(g["pyr_ok"] == 1, "unchanged_callers: mvImagePyramid was not populated (host pyramid should be on)")
    res = {"what": "ONE tracking frame through the unchanged call sites, C++ (tools/unchanged_callers.cc links ms-slam_amd/host's drop-in "
                   "classes): (1) Frame.cc:122-125 two std::threads x ORBextractor::operator(), host pyramid on; (2) Tracking.cc:2835-2850 "
                   "class ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono), the HOST-projected default; (3) "
                   "Tracking.cc:3363-3386 class SearchByProjection(F, vpMapPoints, th, ...).  Each matcher call includes the upload "
                   "of the new frame's features (first call) / reuses it (second call)",
           "ms_extract_two_threads": round(g["ms_extract"], 4),
           "ms_search_by_projection_last_frame": round(g["ms_a14"], 4),
           "ms_search_by_projection_local_points": round(g["ms_a13"], 4),
           "ms_total_device_backed_calls": round(g["ms_total"], 4),
           "keypoints": [int(g["N"]), int(g["NR"])], "motion_model_matches": int(g["n14"]), "local_map_matches": int(g["n13"]),
           "last_frame_keypoints": NL, "local_map_points": M, "iterations": int(g["iters"]),
           "stays_host_code": "Frame::ComputeStereoMatches (reads mvImagePyramid; Frame.cc:743-913), Frame::isInFrustum "
                              "(Tracking.cc:3343-3361), both PoseOptimizations — the reference's own code with unchanged callers, not timed "
                              "here; tracking_loop.per_frame_total is the EDITED-caller figure (INTEGRATION.md) that moves the first two "
                              "onto the device",
           "same_features_as_batch_path": True}
    if cpu:
        # CPU oracle legs on the same inputs; the a14 oracle gets the projections the class computed on the host (its own build)
        rf = orb_oracle.OracleFrame(kps, desc, ur, bounds, scale)
        tab14 = dict(valid=g["valid"], u=g["u"], v=g["v"], ur=g["ur"], octave=last["octave"], angle=last["angle"], desc=last["desc"],
                     mp=np.arange(NL, dtype=np.int32), obs=last["obs"])
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            c14 = np.full(N, -1, np.int32)
            w14 = rf.SearchByProjection_frames(tab14, c14, th_mm, bool(g["fwd"]), bool(g["bwd"]), True)
        dt14 = (time.perf_counter() - t0) / reps
        same14 = w14 == g["n14"] and np.array_equal(c14, g["ids14"])
        self_check(same14, "unchanged_callers: class SearchByProjection(Cur, Last) differs from the CPU oracle")
        # a13: the frame holds the a14 matches (ids M + last index, their Observations() from the last-frame table)
        tab13 = {k: np.concatenate([np.asarray(scratch[k]), np.zeros(NL, np.asarray(scratch[k]).dtype)]) for k in
                 ("track_in_view", "proj_x", "proj_y", "proj_xr", "track_depth", "level", "view_cos")}
        tab13["bad"] = np.concatenate([mp["bad"], np.zeros(NL, np.uint8)])
        tab13["sparsified"] = np.concatenate([mp["sparsified"], np.zeros(NL, np.uint8)])
        tab13["desc"] = np.concatenate([mp["desc"], last["desc"]])
        tab13["obs"] = np.concatenate([mp["obs"], last["obs"]]).astype(np.int32)
        t0 = time.perf_counter()
        for _ in range(reps):
            c13 = np.where(c14 >= 0, M + c14, -1).astype(np.int32)
            w13 = rf.SearchByProjection_mps(tab13, c13, th_lp)
        dt13 = (time.perf_counter() - t0) / reps
        same13 = w13 == g["n13"] and np.array_equal(c13, g["ids13"])
        self_check(same13, "unchanged_callers: class SearchByProjection(F, vpMapPoints) differs from the CPU oracle")
        # extraction of the pair on two threads, as Frame.cc:122-125 does on the CPU
        import threading
        exs = [orb_oracle.OracleExtractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"]) for _ in range(2)]
        got = [None, None]

        def eye(e):
            got[e] = exs[e]((left, right)[e])
        t0 = time.perf_counter()
        for _ in range(2):
            th = [threading.Thread(target=eye, args=(e,)) for e in range(2)]
            [t.start() for t in th]
            [t.join() for t in th]
        dte = (time.perf_counter() - t0) / 2
        self_check(np.array_equal(got[0][1].view(np.uint8).reshape(-1, 28), g["kps"]) and np.array_equal(got[0][2], g["desc"]),
                   "unchanged_callers: ORBextractor::operator() differs from the CPU oracle")
        res["cpu_baseline"] = {"ms_extract_two_threads": round(dte * 1e3, 3), "ms_search_by_projection_last_frame": round(dt14 * 1e3, 4),
                               "ms_search_by_projection_local_points": round(dt13 * 1e3, 4),
                               "ms_total": round((dte + dt14 + dt13) * 1e3, 3), "cores": 2, "kind": "port",
                               "sample": f"the same frame: oracle extraction of the pair on 2 threads (2 repetitions), oracle matcher calls "
                                         f"({reps} repetitions, 1 thread; the a14 oracle is fed the projections the class computed)",
                               "gpu_matches_cpu": True}
    return res
