#!/bin/bash
# usage (GPU box): tools/gpu_pmc.sh <outdir-name> "<counters>" [kernel-substring]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT -o p -- python $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 --isolated > $OUT/bench.log 2>&1
python - <<PY
import csv,collections
rows=list(csv.DictReader(open("$OUT/p_counter_collection.csv")))
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n=r["Kernel_Name"].split("(")[0].replace("void ","").replace("msorb::","")
    if "$3" and "$3" not in n: continue
    d[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n,c in sorted(d.items()):
    print(n, {k: round(sum(v)/len(v)) for k,v in sorted(c.items())})
PY
