"""Build gates on hipcc device listings (csrc/build.sh; which file gets which rules: csrc/flags.sh `ups_file_gates`).

  no-packed-fp32   no v_pk_{add,mul,fma}_f32 anywhere in the listing.  The row-stream kernels' one measured wrong result was a packed fp32
                   add that returned `acc + 0` in lanes 48..63 beside a sibling wave's MFMA section (docs/design/rows_hazard.md); their
                   translation unit is compiled with the target feature off, and this rule is what notices if a toolchain ignores that.
  asm-loads        between an inline-asm `global_load_*` with a REGISTER destination and the next inline-asm `s_waitcnt vmcnt`, nothing
                   reads or writes the destination registers (hipcc does not know they are in flight and has been seen to move them:
                   tools/check_asm_loads.py, whose checker this rule runs on the given listing).

Usage: python tools/check_listing.py --rules "no-packed-fp32 asm-loads" <listing.s>      (exit code 1 when a rule fails)"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    rules = args[args.index("--rules") + 1].split() if "--rules" in args else []
    path = args[-1]
    text = open(path).read()
    rc = 0
    if "no-packed-fp32" in rules:
        hits = [(i + 1, l.strip()) for i, l in enumerate(text.split("\n")) if re.match(r"\s*v_pk_(add|mul|fma)_f32\b", l)]
        print("{}: no-packed-fp32: {} packed fp32 instruction(s)".format(os.path.basename(path), len(hits)))
        for ln, l in hits[:8]:
            print("  line {}: {}".format(ln, l))
        rc |= 1 if hits else 0
    if "asm-loads" in rules:
        import check_asm_loads
        n, bad = check_asm_loads.check_listing(text.split("\n"))
        print("{}: asm-loads: {} inline-asm register loads checked, {} violations".format(os.path.basename(path), n, len(bad)))
        for k, line, l, regs in bad[:20]:
            print("  {} line {}: `{}` touches v{} while its load is in flight".format((k or "?")[:90], line, l, regs))
        rc |= 1 if bad else 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
