"""The end of a training step, queue by queue, from a rocprofv3 --kernel-trace CSV (three-stream run): every kernel that runs in the
last `ms` milliseconds before the step's last Adam launch ends.  Usage: python tools/tail.py <kernel_trace.csv> [ms=4]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows))
adam_ends = [e[1] for e in ev if "adam_kernel" in e[3]]
# the last step's end = the end of the last adam launch; the step before ends at the last adam launch > 10 ms earlier
t_end = adam_ends[-1]
qs = sorted(set(e[2] for e in ev))
sel = [e for e in ev if e[1] > t_end - ms * 1e6 and e[0] <= t_end]
print("queues:", qs)
for s, e, q, n in sel:
    short = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:70]
    print("%9.1f %9.1f us  q%-3s %8.1f us  %s" % ((s - t_end) / 1e3, (e - t_end) / 1e3, qs.index(q), (e - s) / 1e3, short))
