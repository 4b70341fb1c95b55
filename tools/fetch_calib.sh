#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration: tools/fetch_calib.hip under rocprofv3 --pmc (one counter per pass) + plain kernel trace for the bandwidths
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  OUT=$R/gpurun_out/calib_$c; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT -o p -- $R/tools/_fetch_calib > $OUT/cmd.log 2>&1 )
  python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/**/p_counter_collection.csv", recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k,v in d.items(): print("$c", k, "KB=%.0f" % (sum(v)/len(v)), "ratio_to_1GiB=%.3f" % (sum(v)/len(v)*1024/2**30))
t=glob.glob("$OUT/**/p_kernel_trace.csv", recursive=True)[0]
e=collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    e[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in e.items(): print("  (serialized by pmc) dur_us", k, round(min(v),1), "-> %.2f TB/s" % (2**30/min(v)/1e6))
PY
done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/calib_t -o p -- $R/tools/_fetch_calib > /dev/null 2>&1 )
python - <<PY
import csv,glob,collections
t=glob.glob("$R/gpurun_out/calib_t/**/p_kernel_trace.csv", recursive=True)[0]
e=collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    e[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in e.items(): print("dur_us", k, round(min(v),1), "-> %.2f TB/s" % (2**30/min(v)/1e6))
PY
