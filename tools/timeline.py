"""Stream timeline of the training step from a rocprofv3 --kernel-trace CSV: wall time per step, time with at least one
kernel running, time the main queue is idle, and the longest idle gaps with the kernels around them.
Usage: python tools/timeline.py <kernel_trace.csv> [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"][:60]) for r in rows))
# the timed region: the last `steps` occurrences of the optimizer kernel mark step ends
adam = [i for i, e in enumerate(ev) if "adam" in e[3].lower()]
print("kernels", len(ev), "adam launches", len(adam))
t0, t1 = ev[len(ev) // 2][0], ev[-1][1]
sel = [e for e in ev if e[0] >= t0]
wall = t1 - t0
busy = 0; cur_s, cur_e = None, None
gaps = []
prev = None
for s, e, q, n in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, prev, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if prev is None or e >= cur_e: prev = n
busy += cur_e - cur_s
print("window %.2f ms, some kernel running %.2f ms (%.1f%%), idle %.2f ms in %d gaps" % (wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6, len(gaps)))
perq = collections.defaultdict(float)
for s, e, q, n in sel: perq[q] += e - s
for q, v in sorted(perq.items(), key=lambda x: -x[1]): print("  queue %s: kernel time %.2f ms (%.1f%% of window)" % (q, v / 1e6, 100.0 * v / wall))
gaps.sort(reverse=True)
print("largest gaps (us): after -> before")
for g, a, b in gaps[:15]: print("  %8.1f  %s -> %s" % (g / 1e3, a, b))
hist = collections.Counter()
for g, a, b in gaps: hist[min(int(g / 1e3) // 5 * 5, 50)] += g
print("idle time by gap size (us bucket: ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
