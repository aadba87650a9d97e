"""Per-step marks from a rocprofv3 --kernel-trace CSV of bench.py: when (relative to the step's first kernel) the critics' block
starts and ends, when the mask decoder's first kernel of segment B starts, and how long no kernel at all is running inside the step
-- the questions behind an A/B of the critics' launch structure (UPS_TOWERS).
Usage: python tools/probes/step_marks.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# a step starts at the first randn_kernel that follows an adam launch
starts, seen_adam = [], True
for i, (s, e, n) in enumerate(ev):
    if "adam" in n.lower():
        seen_adam = True
    elif "randn_kernel" in n and seen_adam:
        starts.append(i); seen_adam = False
starts = starts[len(starts) // 2:]            # the timed half
out = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = ev[a:b]
    t0 = seg[0][0]
    crit = [(s, e) for s, e, n in seg if "tower_" in n or "critic_head" in n or ("conv_igemm_kernelIDF16bLi64ELi2" in n)]
    cat = [s for s, e, n in seg if "CatArrayBatchedCopy" in n and crit and s > crit[0][0]]
    big = [s for s, e, n in seg if "conv3x3_patch_kernelIDF16_Li128" in n]
    busy, cur_e, idle = 0, seg[0][1], 0
    for s, e, n in seg[1:]:
        if s > cur_e:
            idle += s - cur_e
        cur_e = max(cur_e, e)
    out.append(((ev[b][0] - t0) / 1e6, (crit[0][0] - t0) / 1e6 if crit else -1, (max(e for s, e in crit) - t0) / 1e6 if crit else -1,
                (big[0] - t0) / 1e6 if big else -1, idle / 1e6, len(seg)))
print("step ms | critics' first kernel | critics' last kernel end | first fp16 128-wide patch kernel (mask decoder) | idle ms | launches")
for o in out:
    print("  %7.3f | %7.3f | %7.3f | %7.3f | %6.3f | %d" % o)
if out:
    m = [sum(o[i] for o in out) / len(out) for i in range(6)]
    print("mean %6.3f | %7.3f | %7.3f | %7.3f | %6.3f | %.0f" % tuple(m))
