#!/bin/bash
# per-level durations of the pyramid kernels (kernel trace), for the variants selected by env
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  OUT=$R/gpurun_out/pyr_$v; mkdir -p $OUT
  ( cd $R && env $(echo $v | tr , ' ') rocprofv3 --kernel-trace --output-format csv -d $OUT -o p -- python tools/pyr_levels.py > $OUT/cmd.log 2>&1 )
  tail -1 $OUT/cmd.log
  python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/**/p_kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "pyr_resize" in r["Kernel_Name"]]
d=collections.defaultdict(list)
for r in rows:
    d[(r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Grid_Size_Y"))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items(): print("$v", k, "n=%d"%len(v), "avg_us=%.1f"%(sum(v[len(v)//2:])/len(v[len(v)//2:])))
PY
done
