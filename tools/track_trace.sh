#!/bin/bash
# rocprofv3 kernel trace of the per-frame tracking chain (tools/latency_pair.cc): prints the kernels of the LAST msorb_track_frontend
# call in launch order with start offsets and durations (us) -> gpurun_out/track_trace/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/track_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && LD_LIBRARY_PATH=ms-slam_amd:/opt/rocm/lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- tools/_latency_pair 30 > $OUT/run.log 2>&1 )
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last chain: find the last frame_grid_kernel and print from the preceding pyramid kernel on
idx=[i for i,r in enumerate(rows) if "frame_grid_kernel" in r["Kernel_Name"]]
last=idx[-1]
start=last
while start>0 and int(rows[start]["Start_Timestamp"])-int(rows[start-1]["End_Timestamp"])<60000: start-=1
t0=int(rows[start]["Start_Timestamp"])
for r in rows[start:min(len(rows),last+6)]:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","").replace("msorb::","")[:48]
    print("%-50s +%8.1f us  dur %7.1f us  grid %s" % (n,(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Grid_Size_X"]))
PY
