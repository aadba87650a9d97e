"""Audit of hipcc listings for the hazard that made conv3x3_rows_kernel<bf16,64,6,10> irreproducible (round 5,
docs/design/negative_results.md): an MFMA whose accumulator input SrcC is NOT its destination (hipcc rotates accumulators through
registers: D = A x B + C with D != C) reads SrcC over all of its passes -- the rows of lanes 48..63 last -- and a VALU instruction that
overwrites a register of SrcC a few issue slots later (hipcc reuses the "dead" accumulator for an address computation, behind an
`s_nop 1` of its own) lands before that last read on gfx950: the result's lanes 48..63 of that register are computed from the new value.

For every v_mfma with SrcC != vDst this script scans the following instructions (linear order) for a non-MFMA write to a register
of SrcC and prints the distance in issue slots (s_nop N counts N + 1).  A kernel is SAFE against this when it has no such pair within
`--slots` (default 12) -- e.g. because its accumulators are tied (`mfma(a, b, acc)` assigned back to the same variable in a loop
usually is) or because nothing short-lived is allocated behind them.

Usage: python tools/check_mfma_war.py <listing.s> [--slots N] [--all]
       (listing: hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only file.hip -o listing.s)"""
import re
import sys

path = sys.argv[1]
slots = int(sys.argv[sys.argv.index("--slots") + 1]) if "--slots" in sys.argv else 12
show_all = "--all" in sys.argv
REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs(tok):
    m = REG.search(tok)
    if not m:
        return None
    if m.group(1):
        return (m.group(1), int(m.group(2)), int(m.group(3)))
    return (m.group(4), int(m.group(5)), int(m.group(5)))


def overlap(a, b):
    return a and b and a[0] == b[0] and a[1] <= b[2] and b[1] <= a[2]


funcs, cur, name = [], None, None
for ln, line in enumerate(open(path), 1):
    if re.match(r"^[_A-Za-z][\w$.]*:\s*(;.*)?$", line) and not line.startswith(".L"):
        name = line.split(":")[0]
        cur = []
        funcs.append((name, cur))
        continue
    s = line.strip()
    if cur is None or not s or s[0] in ";." or s.startswith(";;#") or s.endswith(":"):
        continue
    cur.append((ln, s.split(";")[0].strip()))

total = 0
for name, ins in funcs:
    found = []
    for i, (ln, s) in enumerate(ins):
        if not s.startswith("v_mfma") and not s.startswith("v_smfma"):
            continue
        ops = [o.strip() for o in s.split(None, 1)[1].split(",")]
        if len(ops) < 4:
            continue
        d, c = regs(ops[0]), regs(ops[3])
        if c is None or d == c:
            continue
        dist = 0
        for ln2, s2 in ins[i + 1:i + 1 + 4 * slots]:
            op = s2.split()[0]
            if op == "s_endpgm":
                break
            if op == "s_nop":
                dist += int(s2.split()[1]) + 1
                continue
            is_mfma = op.startswith("v_mfma") or op.startswith("v_smfma")
            writes = None
            if (op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_readlane") and not op.startswith("v_readfirstlane")) \
                    or op.startswith("ds_read") or op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("scratch_load"):
                rest = s2.split(None, 1)[1] if " " in s2 else ""
                if "lds" not in op:
                    writes = regs(rest.split(",")[0])
            if writes and overlap(writes, c) and not is_mfma and not op.startswith(("ds_read", "global_load", "buffer_load", "scratch_load")):
                if dist < slots:
                    found.append((ln, s, ln2, s2, dist))
                break
            if writes and overlap(writes, c):
                break           # (an MFMA or a load takes the register over: a different, interlocked or long-latency, case)
            dist += 1
            if dist >= slots:
                break
    if found or show_all:
        print("%s: %d MFMA(s) whose SrcC is overwritten by a VALU within %d issue slots" % (name, len(found), slots))
        for ln, s, ln2, s2, dist in found[:8]:
            print("   line %d  %s\n     -> line %d  %s   (%d slots later)" % (ln, s, ln2, s2, dist))
    total += len(found)
print("total", total)
