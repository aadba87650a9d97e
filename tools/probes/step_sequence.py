"""The kernel sequence of ONE traced step (rocprofv3 --kernel-trace CSV of bench.py): start offset, duration, the gap since anything
was last running, queue, kernel -- to see where the GPU waits for the host.   Usage: python tools/probes/step_sequence.py <csv> [k]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows)
starts, seen_adam = [], True
for i, (s, e, q, n) in enumerate(ev):
    if "adam" in n.lower():
        seen_adam = True
    elif "randn_kernel" in n and seen_adam:
        starts.append(i); seen_adam = False
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 2
a, b = starts[k], starts[k + 1]
t0 = ev[a][0]
cur_e = ev[a - 1][1] if a else t0
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:70]
for s, e, q, n in ev[a:b + 3]:
    gap = max(0, s - cur_e)
    print("%9.1f us  dur %8.1f  gap %7.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, q, short(n)))
    cur_e = max(cur_e, e)
