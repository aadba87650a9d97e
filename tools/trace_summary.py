"""Group a rocprofv3 kernel_trace.csv by (kernel, grid) -> where the time of one kernel family goes."""
import csv, sys, collections
path, pat = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if pat and pat not in n:
        continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = (n.replace("_ZN12_GLOBAL__N_1", "")[:100], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", ""))
    agg[key][0] += 1; agg[key][1] += dur
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print("total us:", round(tot, 1))
for (n, g, w), (c, t) in rows[:40]:
    print(f"{n:100s} grid {g:>9s} wg {w:>4s} calls {c:4d} total {t:10.1f} us avg {t/c:9.1f}")

if len(sys.argv) > 3:   # list individual calls of one grid size
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"] and r.get("Grid_Size_X", r.get("Grid_Size", "")) == sys.argv[3]:
            print(r["Kernel_Name"][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
