#!/usr/bin/env python3
"""Turns one full measurement run in gpurun_out/ (tools/_round_run.sh on the GPU box) into the committed summaries under
profiles/: the bench line, the rocprofv3 --kernel-trace --stats per-kernel table of `bench.py --isolated`, the PMC summary
(FETCH_SIZE / WRITE_SIZE in separate passes, SQ instruction counts), per-frame latencies and the §8f bench."""
import ast, csv, json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "round1"
VALU_LANE_OPS_PER_S = 51.5e12   # measured v_fma_f32 rate 103 TFLOP/s / 2 (MI355X_MICROARCH / cdna_hip_programming guide)


def parse(txt):
    out = {}
    for line in open(txt):
        m = re.match(r"^(.*?) (\{.*\})\s*$", line)
        if m:
            out[m.group(1).strip()] = ast.literal_eval(m.group(2))
    return out


bench = json.loads(open(os.path.join(G, "bench_r1.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(P, f"{rnd}_bench.json"), "w"), indent=1)
shutil.copy(os.path.join(G, "prof_r1g", "r_kernel_stats.csv"), os.path.join(P, f"{rnd}_rocprofv3_kernel_stats_isolated.csv"))
fetch, write, valu = (parse(os.path.join(G, f)) for f in ("pmc_fetch.txt", "pmc_write.txt", "pmc_valu.txt"))
# per-kernel average duration of the same command from the kernel trace (ns)
dur = {}
for r in csv.DictReader(open(os.path.join(G, "prof_r1g", "r_kernel_stats.csv"))):
    n = r["Name"].split("(")[0].replace("void ", "").replace("msorb::", "")
    dur[n] = float(r["AverageNs"])
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
    if k in dur and dur[k] > 0:
        e["valu_lane_ops_per_s"] = round(v["SQ_INSTS_VALU"] * 64 / (dur[k] * 1e-9) / 1e12, 2)
        e["valu_fraction_of_measured_peak"] = round(e["valu_lane_ops_per_s"] * 1e12 / VALU_LANE_OPS_PER_S, 3)
    kern[k] = e
json.dump({"note": "rocprofv3 --pmc, per-launch averages of `bench.py --isolated` (128 stereo pairs = 256 images per launch; the "
                   "pyramid row averages its 7 launches); FETCH_SIZE / WRITE_SIZE in KB as reported, separate passes; "
                   "GRBM_GUI_ACTIVE is summed over the 8 XCDs; valu_lane_ops_per_s = 64 * SQ_INSTS_VALU / average duration of the "
                   "same command's kernel trace; its fraction is against 51.5 T lane-ops/s (the guide's measured 103 TFLOP/s "
                   "v_fma_f32 = the VALU issue rate the chip sustains)", "kernels": kern},
          open(os.path.join(P, f"{rnd}_pmc_summary.json"), "w"), indent=1)
lat = json.load(open(os.path.join(G, "latency_r1.json")))
try:
    lat["cpp_two_threads"] = json.load(open(os.path.join(G, "latency_pair.json")))
    lat["cpp_two_threads"]["note"] = ("tools/latency_pair.cc: the per-frame front-end of the tracking loop (BASELINE configs[2]) in C++ "
                                      "through the C ABI — msorb_extract from two fresh std::threads per frame, one handle per eye, like "
                                      "Frame.cc:122-125, then stereo matching, frame upload and SearchByProjection over 4096 map points; "
                                      "the Python figures above include interpreter and numpy overhead")
except Exception:
    pass
json.dump(lat, open(os.path.join(P, f"{rnd}_latency_per_frame.json"), "w"), indent=1)
shutil.copy(os.path.join(G, "bow_bench.json"), os.path.join(P, f"{rnd}_bow_bench.json"))
if os.path.exists(os.path.join(G, "valu_ubench.txt")):
    shutil.copy(os.path.join(G, "valu_ubench.txt"), os.path.join(P, f"{rnd}_valu_ubench.txt"))
print("value", bench["value"], "ms/step", bench["ms_per_step"], bench["stage_ms_per_step"])
for k, e in kern.items():
    print(f"{k:36s} {e['avg_duration_us']} us  fetch {e['FETCH_SIZE_KB']} KB write {e['WRITE_SIZE_KB']} KB  valu {e.get('valu_fraction_of_measured_peak')}")
