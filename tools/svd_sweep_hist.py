"""Histogram of Jacobi sweeps per SVD call from a TJM_DEBUG_SVD=1 stderr log (diagnostic tool)."""
import collections
import re
import sys

calls = []
cur = None
for line in open(sys.argv[1]):
    m = re.match(r"\[svd\] ncols (\d+) rx (\d+) sweep (\d+) live (\d+) rotations (\d+)", line)
    if not m:
        continue
    nc, rx, sw, live, rot = map(int, m.groups())
    if sw == 0:
        cur = [nc, rx, 0, []]
        calls.append(cur)
    cur[2] = sw + 1
    cur[3].append((live, rot))
h = collections.Counter((c[0], c[1], c[2]) for c in calls)
for k in sorted(h):
    print(k, h[k])
big = [c for c in calls if c[0] >= 256]
if big:
    print("example", big[len(big) // 2])
