#!/bin/bash
# usage (GPU box): tools/pmc_run.sh <outdir-name> "<counters>" <kernel-substring> -- <command...>
# One rocprofv3 --pmc pass (counters + kernel trace only) of <command>; prints per-kernel counter averages.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; CTR=$2; SUB=$3; shift 4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout -k 5 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT -o p -- "$@" > $OUT/cmd.log 2>&1 )
[ -n "$(find $OUT -name p_counter_collection.csv)" ] || { echo "no counters collected: see $OUT/cmd.log"; tail -3 $OUT/cmd.log | cut -c1-300; exit 1; }
python - <<PY
import csv,collections,glob
f=glob.glob("$OUT/**/p_counter_collection.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","").replace("msorb::","")
    if "$SUB" and "$SUB" not in n: continue
    d[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n,c in sorted(d.items()):
    print(n, {k: round(sum(v)/len(v)) for k,v in sorted(c.items())}, "n=%d" % len(next(iter(c.values()))))
PY
