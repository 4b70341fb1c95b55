"""Summary of a rocprofv3 kernel trace of bench.py's two-batches-in-flight loop (tools/timeline.sh writes the trace):
one step's kernels per hardware queue, and how much of the window has 0 / 1 / 2+ kernels running.
usage: python tools/timeline_summary.py gpurun_out/timeline > profiles/roundN_timeline_pipelined.txt"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']),
             r['Kernel_Name'].split('(')[0].split('<')[0].replace('msorb::', '').replace('void ', '')[:26], r.get('Queue_Id', '?')) for r in rows)
d = [i for i, k in enumerate(ks) if k[2].startswith('describe')]
# the pipelined loop = the stretch where describe kernels alternate between two queues
alt = [j for j in range(1, len(d)) if ks[d[j]][3] != ks[d[j - 1]][3]]
if len(alt) < 8:
    sys.exit("no alternating describe kernels found: not a pipelined trace")
j0, j1 = alt[len(alt) // 2 - 2], alt[len(alt) // 2 + 2]
t0, t1 = ks[d[j0]][0], ks[d[j1]][0]
print(f"window: {(t1 - t0) / 1e6:.3f} ms = {j1 - j0} steps under the profiler ({(t1 - t0) / 1e6 / (j1 - j0):.3f} ms per step)")
print("  start_us    end_us   dur_us  queue  kernel")
ev = []
for k in ks:
    if t0 <= k[0] < t1:
        if k[1] - k[0] > 8000:
            print(f"{(k[0]-t0)/1e3:10.1f} {(k[1]-t0)/1e3:9.1f} {(k[1]-k[0])/1e3:8.1f}  q{k[3]}  {'        ' * (int(k[3]) % 4)}{k[2]}")
    if k[1] > t0 and k[0] < t1:
        ev.append((max(k[0], t0), 1)); ev.append((min(k[1], t1), -1))
ev.sort()
busy = {0: 0, 1: 0, 2: 0}
n, last = 0, t0
for t, dn in ev:
    busy[min(n, 2)] += t - last
    n += dn; last = t
busy[min(n, 2)] += t1 - last
tot = float(t1 - t0)
print(f"kernels running: none {busy[0]/tot:.1%}, one {busy[1]/tot:.1%}, two or more {busy[2]/tot:.1%} of the window")
