#!/bin/bash
# kernel trace of msorb_visibility_csr (tools/visibility_bench.cc): the last call's kernels in launch order
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/vis_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && LD_LIBRARY_PATH=ms-slam_amd:/opt/rocm/lib timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- tools/_visibility_bench 20 > $OUT/run.log 2>&1 )
python - <<PY
import csv,glob
rows=list(csv.DictReader(open("$OUT/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "vis_first_kernel" in r["Kernel_Name"]]
a=idx[-1]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:]:
    n=r["Kernel_Name"].split("(")[0].replace("void ","").replace("msorb::","")[:40]
    print("%-42s +%8.1f us  dur %7.1f us  grid %s" % (n,(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Grid_Size_X"]))
for f in glob.glob("$OUT/t_memory_copy_trace.csv"):
    m=list(csv.DictReader(open(f)))
    m.sort(key=lambda r:int(r["Start_Timestamp"]))
    for r in m[-10:]:
        print("copy %-20s +%8.1f us dur %7.1f us" % (r.get("Direction",""),(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
cat $OUT/run.log
