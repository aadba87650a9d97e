"""Instruction histogram of the LOOPS of one kernel in a hipcc -S listing (round-4 verdict, item 4): every backward branch closes a
loop body (label ... s_cbranch to that label); for each loop: MFMA / other VALU / SALU / ds_read / ds_write / vector-memory /
s_waitcnt / s_barrier counts, and the same per 48 MFMAs (one tap-row of the 128-wide patch kernel per wave).
Usage: python tools/asm_loops.py <listing.s> <mangled-name-substring> [min_mfma]"""
import collections
import re
import sys


def kernel_body(lines, key):
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().endswith(":") or (l.startswith("_ZN") and key in l and ": " in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end + 1]


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ds_read"
    if op.startswith("ds_"):
        return "ds_write"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    body = kernel_body(lines, sys.argv[2])
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    ops_all = collections.Counter()
    for l in body:
        t = l.strip()
        if t and not t.startswith((";", ".", "_ZN")) and not t.endswith(":"):
            ops_all[classify(t.split()[0])] += 1
    print("kernel: {} instructions {}".format(sum(ops_all.values()), dict(ops_all)))
    for i, l in enumerate(body):
        t = l.strip()
        m = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)", t) or re.match(r"s_branch (\.LBB\d+_\d+)", t)
        if not m or m.group(1) not in labels or labels[m.group(1)] >= i:
            continue
        lo = labels[m.group(1)]
        cnt = collections.Counter()
        valu = collections.Counter()
        for b in body[lo:i + 1]:
            tt = b.strip()
            if tt and not tt.startswith((";", ".")) and not tt.endswith(":"):
                op = tt.split()[0]
                cnt[classify(op)] += 1
                if classify(op) in ("valu", "salu"):
                    valu[op] += 1
        if cnt["mfma"] < min_mfma:
            continue
        n = cnt["mfma"]
        print("loop {} (lines {}..{}): {}".format(m.group(1), lo, i, dict(cnt)))
        if n == 0:          # a loop without matrix instructions (the part-path kernels): totals only
            print("   most frequent VALU / SALU: " + ", ".join("{} x{}".format(k, v) for k, v in valu.most_common(14)))
            continue
        print("   per MFMA: other VALU {:.2f}  SALU {:.2f}  ds_read {:.2f}  ds_write {:.2f}  vmem {:.2f}  waitcnt {:.2f}  barrier {:.3f}".format(
            cnt["valu"] / n, cnt["salu"] / n, cnt["ds_read"] / n, cnt["ds_write"] / n, cnt["vmem"] / n, cnt["waitcnt"] / n, cnt["barrier"] / n))
        print("   per 48 MFMAs: " + "  ".join("{} {:.1f}".format(k, cnt[k] * 48.0 / n) for k in ("valu", "salu", "ds_read", "ds_write", "vmem", "waitcnt", "barrier")))
        print("   most frequent VALU / SALU: " + ", ".join("{} x{}".format(k, v) for k, v in valu.most_common(14)))


if __name__ == "__main__":
    main()
