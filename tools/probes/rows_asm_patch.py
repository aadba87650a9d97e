"""Listing patcher of the row-stream hazard hunt (tools/asm_patch_build.sh; docs/design/rows_hazard.md).  Reads a hipcc device listing
on stdin, inserts `s_nop` fences at the instruction pairs of ONE hazard class inside the kernels whose mangled name contains --kernel
(default: the irreproducible instance), writes the listing to stdout and the patched sites to stderr.

  modes (comma-separated):
    war_valu   VALU write to a register of the SrcC of an MFMA with D != C issued < W slots earlier          (fence before the VALU)
    war_lds    ds_read / global_load whose DESTINATION overlaps SrcA / SrcB / SrcC of an MFMA < W slots earlier (fence before the load)
    war_mfma   MFMA whose D overlaps SrcC (D != C) of one of the previous two MFMAs                            (fence before the MFMA)
    raw_rot    MFMA with D != C whose SrcC is the D of an MFMA at most two MFMAs earlier                      (fence before the MFMA)
    raw_any    any MFMA whose SrcC is the D of an MFMA at most two MFMAs earlier                              (fence before the MFMA)
    rename     the war_valu sites WITHOUT any fence: the VALU's destination (and its readers up to the next write of that register)
               is renamed to a fresh register above the kernel's allocation (.amdhsa_next_free_vgpr / accum_offset raised by 8):
               the write-after-read pair disappears, the instruction stream and its timing stay exactly as they were
    after_res  --nops wait states behind the `s_waitcnt lgkmcnt(0)` that follows every ds_read_b64 (the epilogue's residual reads):
               delays the FIRST consumer of the returned registers and nothing else
    vmcnt0     `s_waitcnt vmcnt(0)` in front of every s_cbranch that closes the MFMA section: every row DMA has landed before the
               epilogue starts (a test, not a fix: it serialises the DMA latency)
    top        32 nop cycles behind every s_barrier (a pure phase shift: the control)
    tail       32 nop cycles in front of every s_cbranch that closes the MFMA section (the position of -DUPS_ROWS_NOP_ONLY, roughly)
  --insert-after TXT --occurrence K --text "insn; insn"   (mode `insert`) put instructions behind the K-th (1-based) instruction of
               the kernel that contains TXT (mode `replace`: in its place): the observation instrument (copy a suspect register into a neighbouring output channel)
  --rename-from TXT --rename-map "168:208,169:209"   (mode `region`) behind the first instruction containing TXT and up to the end
               of the kernel every mapped VGPR is renamed (single registers and ranges): scratch registers of the epilogue moved off
               the operands of the MFMAs still in flight, instruction for instruction the same stream
  --match TXT  patch only sites whose instruction text contains TXT (alternatives separated by |)
  --nops N     wait states per fence (default 8);  --tailnops N  for the top / tail modes (default 32)
  --window W   look-back in issue slots (default 12)"""
import re
import sys

args = sys.argv[1:]


def opt(name, default):
    return args[args.index(name) + 1] if name in args else default


modes = set(opt("--modes", "war_valu").split(","))
kernel = opt("--kernel", "conv3x3_rows_kernelIDF16bLi64ELi6ELi10ELb0ELi0E")
nops = int(opt("--nops", "8"))
window = int(opt("--window", "12"))
tailnops = int(opt("--tailnops", "32"))
ins_after, ins_occ, ins_text = opt("--insert-after", ""), int(opt("--occurrence", "1")), opt("--text", "")
ins_seen = 0
ren_from = opt("--rename-from", "")
ren_map = dict((int(a), int(b)) for a, b in (kv.split(":") for kv in opt("--rename-map", "").split(",") if kv))
ren_on = False
ren_max = 0


def ren_line(text):
    def one(m):
        return "v%d" % ren_map.get(int(m.group(1)), int(m.group(1)))

    def rng(m):
        a, b = int(m.group(1)), int(m.group(2))
        if a in ren_map or b in ren_map:
            assert all(r in ren_map for r in range(a, b + 1)) and ren_map[b] - ren_map[a] == b - a, "range %s not mapped contiguously" % m.group(0)
            return "v[%d:%d]" % (ren_map[a], ren_map[b])
        return m.group(0)
    text = re.sub(r"\bv\[(\d+):(\d+)\]", rng, text)
    return re.sub(r"\bv(\d+)\b", one, text)
match = opt("--match", "")
REG = re.compile(r"^([va])\[(\d+):(\d+)\]$|^([va])(\d+)$")


def reg(tok):
    m = REG.match(tok.strip())
    if not m:
        return None
    if m.group(1):
        return (m.group(1), int(m.group(2)), int(m.group(3)))
    return (m.group(4), int(m.group(5)), int(m.group(5)))


def overlap(a, b):
    return a is not None and b is not None and a[0] == b[0] and a[1] <= b[2] and b[1] <= a[2]


def fence(n):
    out = []
    while n > 0:
        k = min(n, 16)
        out.append("\ts_nop %d\n" % (k - 1))
        n -= k
    return out


lines = sys.stdin.readlines()
out, sites = [], []
renames = {}       # old register number -> new register number, active until the old register is written again
fresh = [None]     # next fresh register (set from the kernel's .amdhsa_next_free_vgpr on first use)
if "rename" in modes:
    in_k = False
    for l in lines:
        if l.strip().startswith(".amdhsa_kernel"):
            in_k = kernel in l
        if in_k and ".amdhsa_next_free_vgpr" in l:
            fresh[0] = int(l.split()[-1])
    assert fresh[0] is not None, "kernel descriptor not found"
    base_free = fresh[0]
inside = False
hist = []          # (kind, slots_at_issue, operands) of the instructions seen in this function; kind: 'mfma' | 'other'
slot = 0
for ln, line in enumerate(lines, 1):
    s = line.strip()
    if re.match(r"^[_A-Za-z][\w$.]*:\s*(;.*)?$", line) and not line.startswith(".L"):
        inside = kernel in line
        hist, slot = [], 0
        ren_on = False
    if not inside or not s or s[0] in ";." or s.startswith(";;#") or s.endswith(":"):
        out.append(line)
        continue
    code = s.split(";")[0].strip()
    op = code.split()[0]
    ops = [o.strip() for o in code.split(None, 1)[1].split(",")] if " " in code else []
    pre = []
    is_mfma = op.startswith("v_mfma")
    if "region" in modes and ren_on:
        new = ren_line(line)
        if new != line:
            sites.append((ln, "region", new.strip()))
            line = new
            code = line.strip().split(";")[0].strip()
            ops = [o.strip() for o in code.split(None, 1)[1].split(",")] if " " in code else []
    if "region" in modes and ren_from and ren_from in code and not ren_on:
        ren_on = True
    if renames and ops:
        w0 = reg(ops[0]) if (op.startswith(("v_", "ds_read", "global_load")) and not op.startswith(("v_cmp",))) else None
        new_ops = list(ops)
        for i, o in enumerate(ops):
            if i == 0 and w0 is not None:
                continue
            for old, new in renames.items():
                o = re.sub(r"\bv%d\b" % old, "v%d" % new, o)
            new_ops[i] = o
        if new_ops != ops:
            line = "\t" + op + " " + ", ".join(new_ops) + "\n"
            sites.append((ln, "rename-use", line.strip()))
        if w0 is not None:
            for old in [o for o in renames if w0[1] <= o <= w0[2]]:
                del renames[old]
    if op == "s_waitcnt" and "after_res" in modes and "lgkmcnt(0)" in code and hist and hist[-1][3].startswith("ds_read_b64") \
            or (op == "s_waitcnt" and "after_res" in modes and "lgkmcnt(0)" in code and len(hist) > 1 and hist[-2][3].startswith("ds_read_b64")):
        out.append(line)
        out.extend(fence(nops))
        sites.append((ln, "after_res", code))
        slot += 1 + nops
        continue
    if op == "s_nop":
        slot += int(ops[0]) + 1
        out.append(line)
        continue
    mf = [h for h in hist if h[0] == "mfma"]
    if is_mfma and len(ops) >= 4:
        d, a, b, c = reg(ops[0]), reg(ops[1]), reg(ops[2]), reg(ops[3])
        prev2 = mf[-2:]
        if "war_mfma" in modes and any(overlap(d, h[2]["c"]) and h[2]["d"] != h[2]["c"] for h in prev2):
            pre, why = fence(nops), "war_mfma"
        if "raw_rot" in modes and d != c and any(h[2]["d"] == c for h in prev2):
            pre, why = fence(nops), "raw_rot"
        if "raw_any" in modes and any(h[2]["d"] == c for h in prev2):
            pre, why = fence(nops), "raw_any"
    elif op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and ops:
        w = reg(ops[0])
        if "war_valu" in modes:
            for h in mf:
                if slot - h[1] < window and h[2]["d"] != h[2]["c"] and overlap(w, h[2]["c"]):
                    pre, why = fence(nops), "war_valu"
        if "rename" in modes and w is not None and w[1] == w[2] and (not match or any(m in code for m in match.split("|"))):
            for h in mf:
                if slot - h[1] < window and h[2]["d"] != h[2]["c"] and overlap(w, h[2]["c"]):
                    renames[w[1]] = fresh[0]
                    line = re.sub(r"^(\s*\S+\s+)v%d\b" % w[1], r"\g<1>v%d" % fresh[0], line)
                    sites.append((ln, "rename-def", line.strip()))
                    fresh[0] += 1
                    break
    elif op.startswith(("ds_read", "global_load", "buffer_load")) and "lds" not in op and ops:
        w = reg(ops[0])
        if "war_lds" in modes:
            for h in mf:
                if slot - h[1] < window and (overlap(w, h[2]["c"]) or overlap(w, h[2]["a"]) or overlap(w, h[2]["b"])):
                    pre, why = fence(nops), "war_lds"
    elif op == "s_barrier" and "top" in modes:
        out.append(line)
        out.extend(fence(tailnops))
        sites.append((ln, "top", code))
        slot += 1
        continue
    elif op.startswith("s_cbranch") and "tail" in modes and mf and slot - mf[-1][1] <= 2:
        pre, why = fence(tailnops), "tail"
    elif op.startswith("s_cbranch") and "vmcnt0" in modes and mf and slot - mf[-1][1] <= 2:
        out.append("\ts_waitcnt vmcnt(0)\n")
        sites.append((ln, "vmcnt0", code))
    if pre and match and not any(m in code for m in match.split("|")):
        pre = []
    if pre:
        sites.append((ln, why, code))
        out.extend(pre)
        slot += sum(int(p.split()[1]) + 1 for p in pre)
    out.append(line)
    if "replace" in modes and ins_after and ins_after in code:
        ins_seen += 1
        if ins_seen == ins_occ:
            out.pop()
            for t in ins_text.split(";"):
                out.append("\t" + t.strip() + "\n")
            sites.append((ln, "replace", code + "  -> " + ins_text))
    if "insert" in modes and ins_after and ins_after in code:
        ins_seen += 1
        if ins_seen == ins_occ:
            for t in ins_text.split(";"):
                out.append("\t" + t.strip() + "\n")
            sites.append((ln, "insert", code + "  ++ " + ins_text))
    if is_mfma and len(ops) >= 4:
        hist.append(("mfma", slot, {"d": reg(ops[0]), "a": reg(ops[1]), "b": reg(ops[2]), "c": reg(ops[3])}, code))
    else:
        hist.append(("other", slot, None, code))
    slot += 1
    if op.startswith(("s_cbranch", "s_branch")) and "tail" not in modes:
        pass          # (linear scan across branches: a look-back over a never-taken branch is what the hardware sees, too)
if "region" in modes:
    in_k = False
    need = (max(ren_map.values()) + 8) // 8 * 8
    for i, l in enumerate(out):
        if l.strip().startswith(".amdhsa_kernel"):
            in_k = kernel in l
        if in_k and (".amdhsa_next_free_vgpr" in l or ".amdhsa_accum_offset" in l):
            out[i] = re.sub(r"\d+\s*$", str(max(need, int(l.split()[-1]))) + "\n", l)
if "rename" in modes:
    assert fresh[0] - base_free <= 8
    in_k = False
    for i, l in enumerate(out):
        if l.strip().startswith(".amdhsa_kernel"):
            in_k = kernel in l
        if in_k and (".amdhsa_next_free_vgpr" in l or ".amdhsa_accum_offset" in l):
            out[i] = l.replace(str(base_free), str(base_free + 8))
sys.stdout.writelines(out)
for ln, why, code in sites:
    sys.stderr.write("  patched line %d (%s): %s\n" % (ln, why, code))
sys.stderr.write("%d site(s) patched in kernels matching %r, modes %s, %d wait states each\n" % (len(sites), kernel, sorted(modes), nops))
