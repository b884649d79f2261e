import collections, re, sys
calls=[]; cur=None
for line in open(sys.argv[1], errors='ignore'):
    m=re.match(r"\[svd\] ncols (\d+) rx (\d+) sweep (\d+) live (\d+) rotations (\d+)", line)
    if not m: continue
    nc,rx,sw,live,rot=map(int,m.groups())
    if sw==0:
        cur=[nc,rx,[]]; calls.append(cur)
    cur[2].append((live,rot))
h=collections.Counter((c[0],c[1],len(c[2])) for c in calls)
for k in sorted(h): print(k,h[k])
# rotations of each sweep for the (512,1024) shifts
sh=[c for c in calls if c[0]==512 and c[1]==1024]
print('shift solves', len(sh))
import statistics
for s in range(4):
    v=[c[2][s][1] for c in sh if len(c[2])>s]
    if v: print('sweep',s,'n',len(v),'median rot',statistics.median(v),'max',max(v),'zero frac',sum(1 for x in v if x==0)/len(v))
print('examples', [c[2] for c in sh[100:106]])
