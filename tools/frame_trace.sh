#!/bin/bash
# rocprofv3 kernel + memory-copy + HIP-runtime trace of tools/frame_chain.py: the LAST tracking frame (one msorb_track_frontend_motion
# + one msorb_search_local_points) as a merged timeline -> stdout; raw CSVs under gpurun_out/frame_trace/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/frame_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $OUT -o t -- python tools/frame_chain.py 40 > $OUT/run.log 2>&1 )
tail -1 $OUT/run.log
python - <<PY
import csv, glob, os
def load(pat):
    f = glob.glob(os.path.join("$OUT", "**", pat), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
ev = []
for r in load("t_kernel_trace.csv"):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("msorb::", "").replace("(anonymous namespace)::", "")[:46]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", n + "  grid " + r["Grid_Size_X"]))
for r in load("t_memory_copy_trace.csv"):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "") + "  " + r.get("Bytes", r.get("Size", "")) + " B"))
api = []
for r in load("t_hip_api_trace.csv"):
    api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "A", r["Function"]))
ev.sort()
# frames: the extract_stereo_frame-only loop runs last; the frame_total loop before it.  Take the last chain that contains last_frame_kernel.
idx = [i for i, e in enumerate(ev) if "last_frame_kernel" in e[3]]
if not idx:
    raise SystemExit("no last_frame_kernel in the trace")
last = idx[-1]
start = last
while start > 0 and ev[start][0] - ev[start - 1][1] < 80000: start -= 1
end = last
while end + 1 < len(ev) and ev[end + 1][0] - ev[end][1] < 80000 and "pyr_resize" not in ev[end + 1][3]: end += 1
t0 = ev[start][0]
print("-- device timeline of the last tracking frame (us from the first event)")
for s, e, k, n in ev[start:end + 1]:
    print("%s %-62s +%8.1f  dur %7.1f" % (k, n, (s - t0) / 1e3, (e - s) / 1e3))
print("-- host API calls in the same window (>= 4 us, or synchronising)")
for s, e, k, n in sorted(api):
    if s >= t0 - 150000 and s <= ev[end][1] and ((e - s) >= 4000 or "Synchronize" in n):
        print("A %-62s +%8.1f  dur %7.1f" % (n, (s - t0) / 1e3, (e - s) / 1e3))
PY
