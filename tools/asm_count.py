"""Instruction mix of one kernel in a hipcc -S listing.  Usage: python tools/asm_count.py <listing.s> <name-substring> ..."""
import collections, re, sys
s = open(sys.argv[1]).read()
for key in sys.argv[2:]:
    m = re.search(r'^(_ZN\S*' + re.escape(key) + r'\S*):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M)
    if not m:
        print(key, "not found"); continue
    body = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    kinds = collections.Counter(l.split()[0].split('_')[0] for l in body)
    valu = collections.Counter(l.split()[0] for l in body if l.startswith('v_'))
    print(key, "total", len(body), dict(kinds))
    print("   VALU:", valu.most_common(16))
