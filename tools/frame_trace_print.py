#!/usr/bin/env python3
"""Merged timeline of the last tracking frame from a rocprofv3 trace directory (tools/frame_trace.sh)."""
import csv
import glob
import os
import sys

OUT = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/frame_trace"


def load(pat):
    f = glob.glob(os.path.join(OUT, "**", pat), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def kname(s):
    s = s.replace("(anonymous namespace)::", "").replace("void ", "").replace("msorb::", "")
    return s.split("(")[0][:46]


ev = []
for r in load("t_kernel_trace.csv"):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", kname(r["Kernel_Name"]) + "  grid " + r["Grid_Size_X"]))
for r in load("t_memory_copy_trace.csv"):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "") + "  " + r.get("Bytes", r.get("Size", "")) + " B"))
api = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "A", r["Function"]) for r in load("t_hip_api_trace.csv")]
ev.sort()
# anchor: the frame grid launch of the last tracking frame (since round 5 it carries the motion-model projection's blocks too)
kern = [i for i, e in enumerate(ev) if e[2] == "K" and "copyBuffer" not in e[3]]
idx = [kern[j] for j in range(1, len(kern)) if "window_topk" in ev[kern[j]][3] and ("frame_grid_kernel" in ev[kern[j - 1]][3] or "last_frame_kernel" in ev[kern[j - 1]][3])]
if not idx:
    raise SystemExit("no motion-model search (frame_grid_kernel -> window_topk_kernel) in the trace")
last = idx[-1]
start = last
while start > 0 and ev[start][0] - ev[start - 1][1] < 45000:
    start -= 1
end = last
while end + 1 < len(ev) and ev[end + 1][0] - ev[end][1] < 45000 and "pyr_resize" not in ev[end + 1][3] and "pyr_tower" not in ev[end + 1][3]:
    end += 1
t0 = ev[start][0]
print("-- device timeline of the last tracking frame (us from the first event)")
for s, e, k, n in ev[start:end + 1]:
    print("%s %-62s +%8.1f  dur %7.1f" % (k, n, (s - t0) / 1e3, (e - s) / 1e3))
print("-- host API calls in the same window (>= 3 us, or synchronising)")
for s, e, k, n in sorted(api):
    if t0 - 120000 <= s <= ev[end][1] and ((e - s) >= 3000 or "Synchronize" in n):
        print("A %-62s +%8.1f  dur %7.1f" % (n, (s - t0) / 1e3, (e - s) / 1e3))
