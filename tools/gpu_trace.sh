#!/bin/bash
# usage (on the GPU box, via gpurun): tools/gpu_trace.sh <outdir-name> [bench args...]
# rocprofv3 kernel trace of bench.py; prints per-kernel (name, grid) average durations in microseconds.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python $R/bench.py --steps 3 --warmup 1 --lean --isolated "$@" > $OUT/bench.log 2>&1
python - <<PY
import csv,collections
rows=list(csv.DictReader(open("$OUT/r_kernel_trace.csv")))
d=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","").replace("msorb::","")
    d[(n,r["Grid_Size_X"],r["Grid_Size_Y"],r["Grid_Size_Z"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items()):
    print("%-40s grid=%-22s n=%-3d avg_us=%.1f" % (k[0], "x".join(k[1:]), len(v), sum(v)/len(v)/1000))
PY
tail -1 $OUT/bench.log | cut -c1-600
