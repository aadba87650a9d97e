"""Static guard for the hand-counted register loads of the row-stream kernels (conv3x3_rows.hip).

Those kernels issue `global_load_*` from INLINE ASM so that hipcc's wait-count pass does not see them (a load it sees gets a
`vmcnt` wait that ignores the LDS-DMA requests in flight and drains the row pipeline), and wait for them with their own counted
`s_waitcnt` -- also inline asm, with the loaded registers as "+v" operands.  What the compiler does NOT know is that those registers
are not valid between the two statements: when register pressure tells it to, it moves them (`v_mov_b64 v[234:235], v[182:183]`)
BEFORE the wait and the kernel silently multiplies by stale data (round 5: one more conditional store in conv3x3_rows2_kernel did
exactly that; 0.1 % of the elements wrong, not reproducible).

This script compiles the file to assembly and checks, for every kernel: between an asm-block `global_load_{ubyte,dword,dwordx2,...}`
and the next asm-block `s_waitcnt vmcnt`, no instruction reads or writes the load's destination registers.
Round 6: a BUILD GATE (csrc/build.sh -> tools/check_listing.py, rule `asm-loads`).  The forms that violated it (the two-tile input
gradient, the DG == 2 one-tile form) are deleted; what is left to check is conv3x3_rows_maskgrad_kernel's view loads.
Usage: python tools/check_asm_loads.py [file.hip ...]      (exit code 1 on a violation)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unsupervised-part-segmentation_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def regs_of(operand):
    m = re.match(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", operand)
    return {int(m.group(1))} if m else set()


def regs_in(line):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", line):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", line):
        out.add(int(m.group(1)))
    return out


def check(path):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-inline-asm", "-S",
                               "--cuda-device-only", os.path.abspath(path), "-o", out], cwd=os.path.dirname(os.path.abspath(path)), stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    return check_listing(lines)


def check_listing(lines):
    """The check itself, on the lines of a device listing (the build gate tools/check_listing.py runs it on the listing of the very
    flags the object was built with)."""
    bad, kernel, in_asm, pending, n_loads = [], None, False, {}, 0
    for i, raw in enumerate(lines):
        ln = raw.strip()
        if raw.startswith("_Z") and ":" in raw and not raw.startswith("_ZN") is False or (raw.startswith("_Z") and raw.split(":")[0] == raw.split()[0].rstrip(":")):
            kernel, pending = raw.split(":")[0], {}
            continue
        if ln.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if ln.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not ln or ln.startswith((";", ".")) or ln.endswith(":"):
            continue
        op = ln.split()[0]
        if in_asm and re.match(r"global_load_(ubyte|sbyte|ushort|sshort|dword|dwordx2|dwordx3|dwordx4)$", op):
            dst = ln.split()[1].rstrip(",")
            for r in regs_of(dst):
                pending[r] = i + 1
            n_loads += 1
            continue
        if in_asm and op == "s_waitcnt" and "vmcnt" in ln:
            pending = {}
            continue
        if op == "s_endpgm":
            pending = {}
            continue
        if pending:
            hit = regs_in(ln) & set(pending)
            if hit:
                bad.append((kernel, i + 1, ln, sorted(hit)))
    return n_loads, bad


def main():
    files = sys.argv[1:] or [os.path.join(CSRC, "conv3x3_rows.hip")]
    rc = 0
    for f in files:
        n, bad = check(f)
        print("{}: {} inline-asm register loads checked, {} violations".format(os.path.basename(f), n, len(bad)))
        for k, line, text, regs in bad[:20]:
            print("  {} line {}: `{}` touches v{} while its load is in flight".format((k or "?")[:90], line, text, regs))
        rc |= 1 if bad else 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
