"""Slice one kernel out of a hipcc -S listing and summarise it: scratch (spill) instructions and where they sit relative to
the MFMA span.  Usage: python tools/asm_fn.py <listing.s> <mangled-name-substring> [--dump]"""
import sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l and l.rstrip().split(':')[0].endswith('iiiii') or (l.startswith('_ZN') and key in l and ': ' in l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
sc = [i for i, l in enumerate(body) if 'scratch_' in l]
print("lines", len(body), "mfma", len(mf), "span", (mf[0], mf[-1]) if mf else None, "scratch ops", len(sc))
print("scratch inside mfma span:", sum(1 for i in sc if mf and mf[0] <= i <= mf[-1]), "before:", sum(1 for i in sc if mf and i < mf[0]), "after:", sum(1 for i in sc if mf and i > mf[-1]))
if '--dump' in sys.argv:
    print('\n'.join(body))
