"""Per-queue kernel totals from a rocprofv3 --kernel-trace CSV: for the busiest queues, time per (kernel, grid) group per step.
Usage: python tools/by_queue.py <kernel_trace.csv> <steps> [top]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
# the timed region: the second half of the trace
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
perq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
qtot = collections.Counter()
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"])[:58]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")))
    e = perq[r["Queue_Id"]][key]
    e[0] += 1; e[1] += d
    qtot[r["Queue_Id"]] += d
for q, t in qtot.most_common(4):
    print("== queue {}: {:.2f} ms of kernels per step".format(q, t / 1e3 / steps))
    for (name, grid), (n, us) in sorted(perq[q].items(), key=lambda kv: -kv[1][1])[:top]:
        print("  {:7.1f} us/step {:6.1f} x {:8.1f} us  grid {:>9}  {}".format(us / steps, n / steps, us / n, grid, name))
