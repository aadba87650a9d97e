"""Per-kernel counts of what a hot loop should not contain, from the device listings csrc/build.sh leaves in csrc/build/*.s:
full drains of the vector-memory counter (`s_waitcnt vmcnt(0)`: hipcc puts one behind every load that sits under a branch), global
loads / stores and 64-bit vector multiplies (address arithmetic that did not stay on the scalar unit).  A diagnostic, not a gate: cold
paths (per-element fall-back epilogues, table paths of border rows) legitimately hold many of each.  Round 6 found the first-layer
kernel's serialised fragment loads and the patch kernel's per-column CoordConv table path with it.
Usage: python tools/scan_listings.py [min_vmcnt0] [csrc/build]"""
import glob
import os
import re
import sys


def kernels(path):
    cur, body = None, []
    for ln in open(path).read().split("\n"):
        m = re.match(r"^(_Z\S+):", ln)
        if m and not ln.startswith("\t"):
            cur, body = m.group(1), []
        elif "s_endpgm" in ln and cur is not None:
            yield cur, "\n".join(body)
            cur, body = None, []
        elif cur is not None:
            body.append(ln)


def main():
    thr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    root = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                            "unsupervised-part-segmentation_amd", "csrc", "build")
    rows = []
    for f in sorted(glob.glob(os.path.join(root, "*.s"))):
        for name, b in kernels(f):
            w0 = len(re.findall(r"s_waitcnt vmcnt\(0\)", b))
            if w0 < thr:
                continue
            rows.append((w0, len(re.findall(r"global_load_|buffer_load_", b)), len(re.findall(r"global_store_", b)),
                         len(re.findall(r"v_mad_[iu]64|v_mul_lo_u32|v_mul_hi_u32", b)), b.count("\n"), os.path.basename(f), name))
    rows.sort(reverse=True)
    print("{:>8s} {:>6s} {:>6s} {:>6s} {:>7s}  {:22s} kernel".format("vmcnt(0)", "loads", "stores", "mul64", "lines", "file"))
    for r in rows:
        print("{:8d} {:6d} {:6d} {:6d} {:7d}  {:22s} {}".format(*r[:6], r[6][:110]))


if __name__ == "__main__":
    main()
