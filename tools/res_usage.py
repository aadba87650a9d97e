"""Per-kernel VGPRs / spills / scratch / LDS / occupancy from a `hipcc -Rpass-analysis=kernel-resource-usage` log.
Usage: python tools/res_usage.py <log> [name-filter]   (lines: vgprs agprs spill scratch lds occupancy name)
       python tools/res_usage.py <log> --patch [f8]    conv3x3_patch_kernel instances with their template arguments decoded
                                                       (T BN OCC SUB F8 PRE TAPS DMAP; `f8`: the fp8 instances only)"""
import re
import subprocess
import sys


def parse(path):
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return rows


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return out[:len(names)]
    except Exception:
        return names


def patch_table(rows, only_f8):
    """The conv3x3_patch_kernel instances, template arguments read off the mangled name (what tools/resusage.py printed)."""
    for r in rows:
        m = re.search(r"conv3x3_patch_kernelI(\w+?)Li(\d+)ELi(\d)ELi(\d+)ELi(\d)ELb(\d)ELi(\d)ELb(\d)", r["name"])
        if not m or (only_f8 and m.group(5) == "0"):
            continue
        print("T=%s BN=%s OCC=%s SUB=%s F8=%s PRE=%s TAPS=%s DMAP=%s" % m.groups(), "vgpr", r.get("vgpr", -1), "agpr", r.get("agpr", 0),
              "spill", r.get("spill", 0), "scratch", r.get("scratch", 0), "lds", r.get("lds", 0))


if __name__ == "__main__":
    rows = parse(sys.argv[1])
    if len(sys.argv) > 2 and sys.argv[2] == "--patch":
        patch_table(rows, len(sys.argv) > 3)
        sys.exit(0)
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        if flt in n:
            print("{:4d} {:4d} {:5d} {:5d} {:7d} {:2d}  {}".format(r.get("vgpr", -1), r.get("agpr", 0), r.get("spill", 0), r.get("scratch", 0), r.get("lds", 0),
                                                               r.get("occ", 0), n[:150]))
