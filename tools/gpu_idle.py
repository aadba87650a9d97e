"""GPU idle fraction and concurrency of the steady-state training steps from a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# steady state: last 60% of the trace
t0 = rows[0][0]; t1 = max(r[1] for r in rows)
lo = t0 + int(0.4 * (t1 - t0))
rows = [r for r in rows if r[0] >= lo]
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
gaps = []
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = max(r[1] for r in rows) - rows[0][0]
tot = sum(e - s for s, e, _ in rows)
print("window {:.1f} ms, busy (union) {:.1f} ms = {:.1f}%, sum of kernel durations {:.1f} ms (avg concurrency {:.2f})".format(
    wall / 1e6, busy / 1e6, 100.0 * busy / wall, tot / 1e6, tot / busy))
gaps.sort(reverse=True)
print("largest gaps (us, next kernel):")
for g, n in gaps[:12]:
    print("  {:8.1f}  {}".format(g / 1e3, n[:90]))
print("gaps > 5us: {} totalling {:.2f} ms; all gaps {:.2f} ms".format(sum(1 for g, _ in gaps if g > 5000), sum(g for g, _ in gaps if g > 5000) / 1e6, sum(g for g, _ in gaps) / 1e6))
