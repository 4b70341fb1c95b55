#!/usr/bin/env python3
"""Turns one full measurement run in gpurun_out/ (tools/round_run.sh on the GPU box) into the committed summaries under
profiles/: the bench line, the rocprofv3 --kernel-trace --stats per-kernel table of `bench.py --isolated`, the PMC summary
(FETCH_SIZE / WRITE_SIZE in separate passes, SQ instruction counts), per-frame latencies and the §8f bench."""
import ast, csv, json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "round5"
VALU_LANE_OPS_PER_S = 51.5e12   # measured v_fma_f32 rate 103 TFLOP/s / 2 (MI355X_MICROARCH / cdna_hip_programming guide)


def parse(txt):
    out = {}
    for line in open(txt):
        m = re.match(r"^(.*?) (\{.*\})(?:\s+n=\d+)?\s*$", line)
        if m:
            out[m.group(1).strip()] = ast.literal_eval(m.group(2))
    return out


G = os.path.join(G, "round")
bench = json.loads(open(os.path.join(G, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(P, f"{rnd}_bench.json"), "w"), indent=1)
shutil.copy(os.path.join(G, "trace", "r_kernel_stats.csv"), os.path.join(P, f"{rnd}_rocprofv3_kernel_stats_isolated.csv"))
fetch, write, valu = (parse(os.path.join(G, f)) for f in ("pmc_fetch.txt", "pmc_write.txt", "pmc_valu.txt"))
# per-kernel average duration of the same command from the kernel trace (ns)
dur = {}
for r in csv.DictReader(open(os.path.join(G, "trace", "r_kernel_stats.csv"))):
    n = r["Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ", "").replace("msorb::", "")
    dur[n] = float(r["AverageNs"])
# FETCH_SIZE calibration of this run (tools/fetch_calib.hip): reported KB * 1024 / bytes really read
calib = {}
for line in open(os.path.join(G, "fetch_calib.txt")):
    m = re.match(r"^(FETCH_SIZE|WRITE_SIZE) (\S+) KB=(\d+) ratio_to_1GiB=([\d.]+)", line)
    if m and float(m.group(4)) > 0.01:
        calib[f"{m.group(1)}:{m.group(2)}"] = float(m.group(4))
    m = re.match(r"^dur_us (\S+) ([\d.]+) -> ([\d.]+) TB/s", line)
    if m:
        calib[f"TB/s:{m.group(1)}"] = float(m.group(3))
fetch_scale = 1.0 / calib.get("FETCH_SIZE:calib_read_dword", 0.5)
kern = {}
for k in sorted(valu):
    if k.startswith("at::") or k.startswith("__amd"):
        continue
    v = valu[k]
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    e = dict(FETCH_SIZE_KB=fetch.get(k, {}).get("FETCH_SIZE"), WRITE_SIZE_KB=write.get(k, {}).get("WRITE_SIZE"),
             SQ_WAVES=v["SQ_WAVES"], SQ_INSTS_VALU=v["SQ_INSTS_VALU"], SQ_INSTS_SALU=v["SQ_INSTS_SALU"],
             SQ_INSTS_LDS=v["SQ_INSTS_LDS"], GRBM_GUI_ACTIVE=v["GRBM_GUI_ACTIVE"], kernel_cycles_per_xcd=cyc,
             avg_duration_us=round(dur.get(k, 0) / 1e3, 1) if k in dur else None)
    if e["FETCH_SIZE_KB"] is not None and e["WRITE_SIZE_KB"] is not None:
        e["hbm_bytes_calibrated"] = int((e["FETCH_SIZE_KB"] * fetch_scale + e["WRITE_SIZE_KB"]) * 1024)
    if k in dur and dur[k] > 0:
        e["valu_lane_ops_per_s"] = round(v["SQ_INSTS_VALU"] * 64 / (dur[k] * 1e-9) / 1e12, 2)
        e["valu_fraction_of_measured_peak"] = round(e["valu_lane_ops_per_s"] * 1e12 / VALU_LANE_OPS_PER_S, 3)
    kern[k] = e
json.dump({"note": "rocprofv3 --pmc, per-launch averages of `bench.py --isolated` (128 stereo pairs = 256 images per launch; the "
                   "pyramid row averages its 7 launches); FETCH_SIZE / WRITE_SIZE in KB as reported, separate passes; "
                   "hbm_bytes_calibrated = (FETCH_SIZE_KB * fetch_scale + WRITE_SIZE_KB) * 1024 with fetch_scale from the calibration "
                   "kernels of the same run (FETCH_SIZE reports half of the bytes a coalesced 4 B/lane or 16 B/lane stream reads; "
                   "WRITE_SIZE is exact); GRBM_GUI_ACTIVE is summed over the 8 XCDs; valu_lane_ops_per_s = 64 * SQ_INSTS_VALU / "
                   "average duration of the same command's kernel trace; its fraction is against 51.5 T lane-ops/s (the guide's "
                   "measured 103 TFLOP/s v_fma_f32 = the VALU issue rate the chip sustains)",
           "fetch_scale": fetch_scale, "calibration": calib, "kernels": kern},
          open(os.path.join(P, f"{rnd}_pmc_summary.json"), "w"), indent=1)
lat = json.load(open(os.path.join(G, "latency.json")))
try:
    lat["cpp_two_threads"] = json.load(open(os.path.join(G, "latency_pair.json")))
    lat["cpp_two_threads"]["note"] = ("tools/latency_pair.cc: the per-frame front-end of the tracking loop (BASELINE configs[2]) in C++ "
                                      "through the C ABI — msorb_extract from two fresh std::threads per frame, one handle per eye, like "
                                      "Frame.cc:122-125, then stereo matching, frame upload and SearchByProjection over 4096 map points; "
                                      "the Python figures above include interpreter and numpy overhead")
    lat["drop_in_class"] = {"host_pyramid_on": json.load(open(os.path.join(G, "latency_class_on.json"))),
                            "host_pyramid_off": json.load(open(os.path.join(G, "latency_class_off.json"))),
                            "note": "tools/latency_class.cc: the same pair of eye threads through ORB_SLAM3::ORBextractor::operator() "
                                    "(ms-slam_amd/host), with and without the mvImagePyramid read-back"}
except Exception as ex:
    print("latency extras missing:", ex)
json.dump(lat, open(os.path.join(P, f"{rnd}_latency_per_frame.json"), "w"), indent=1)
for src, dst in (("bow_bench.json", "bow_bench.json"), ("valu_ubench.txt", "valu_ubench.txt"), ("valu_ubench2.txt", "valu_ubench2.txt"), ("kf_store_bench.json", "kf_store_bench.json"),
                 ("bench_unique128.json", "bench_unique128.json"), ("bench_unique8_same_steps.json", "bench_unique8_same_steps.json"), ("fetch_calib.txt", "fetch_calib.txt"),
                 ("hamming_bench.txt", "hamming_bench.txt"), ("pytest_gpu.txt", "pytest_gpu.txt"),
                 ("batch_sweep.jsonl", "batch_sweep.jsonl"), ("visibility_bench.json", "visibility_bench.json"),
                 ("track_trace.txt", "tracking_chain_kernel_trace.txt"), ("timeline_pipelined.txt", "timeline_pipelined.txt"),
                 ("frame_chain.json", "frame_chain.json"), ("frame_trace.txt", "frame_trace.txt"), ("qt_marks.txt", "quadtree_phase_marks.txt"),
                 ("describe_pmc.txt", "describe_pmc_raw.txt"), ("frame_copies_ab.txt", "frame_copies_ab.txt"),
                 ("frame_paths_tower_ab.txt", "frame_paths_tower_ab.txt"), ("batch_paths_ab.txt", "batch_paths_ab.txt"), ("per_frame_classes.txt", "per_frame_classes.txt"),
                 ("fast_pmc.txt", "fast_pmc.txt"), ("describe_order_ab.txt", "describe_order_ab.txt")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, f"{rnd}_{dst}"))
print("value", bench["value"], "ms/step", bench["ms_per_step"], bench["stage_ms_per_step"])
for k, e in kern.items():
    print(f"{k:36s} {e['avg_duration_us']} us  fetch {e['FETCH_SIZE_KB']} KB write {e['WRITE_SIZE_KB']} KB  valu {e.get('valu_fraction_of_measured_peak')}")
