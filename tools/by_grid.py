"""Per (kernel, grid) totals of a rocprofv3 --kernel-trace CSV: which launches of a shape-generic kernel carry the time.
Usage: python tools/by_grid.py <kernel_trace.csv> <steps> [min_ms_per_step]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = (r["Kernel_Name"][:96], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r.get("Grid_Size_Y", 1)))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[k][0] += 1; agg[k][1] += d
tot = sum(v[1] for v in agg.values())
print("total kernel time %.2f ms/step over %d launches/step" % (tot / 1e6 / steps, len(rows) / steps))
for k, v in sorted(agg.items(), key=lambda x: -x[1][1]):
    ms = v[1] / 1e6 / steps
    if ms < floor: break
    print("%-96s blocks %7d y %4d  %5.1f launches/step  %7.1f us avg  %6.2f ms/step" % (k[0], k[1], k[2], v[0] / steps, v[1] / v[0] / 1e3, ms))
