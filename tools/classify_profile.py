"""Summarise tools/layer_profile.py output by conv class (3x3 wide / thin / stride 2 / small images / 1x1)."""
import collections, sys
rows = []
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) == 10 and p[1] in ('fwd', 'dgrad', 'wgrad'):
        rows.append((p[0], p[1], p[3], int(p[4]), int(p[5]), int(p[6]), int(p[7]), float(p[8]), float(p[9])))
    elif l.startswith("step"):
        print(l.strip())
cls = collections.defaultdict(float)
for name, ps, x, k, s, cin, co, ms, tf in rows:
    n, h, w, c = map(int, x.split('x'))
    if k == 1: key = '1x1'
    elif s == 2: key = '3x3s2'
    elif h % 16: key = '3x3 small'
    elif min(cin, co) >= 128: key = '3x3 wide'
    else: key = '3x3 thin'
    cls[(key, ps)] += ms
for key in sorted(set(k for k, _ in cls)):
    print("{:10s} fwd {:6.2f} dgrad {:6.2f} wgrad {:6.2f}  total {:6.2f}".format(
        key, cls[(key, 'fwd')], cls[(key, 'dgrad')], cls[(key, 'wgrad')], sum(cls[(key, q)] for q in ('fwd', 'dgrad', 'wgrad'))))
