"""Reads a rocprofv3 kernel trace of bench.py and reports, for one step (between two K8 launches), how long
the GPU ran 0, 1, 2.. blend kernels at once and when each stream was active.
    python tools/trace_concurrency.py <..._kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Stream_Id']) for r in rows)
pb = [e for e in ev if 'param_bwd' in e[2]]
a, b = pb[-3], pb[-2]
t0, t1 = a[1], b[1]
inside = [e for e in ev if e[0] >= t0 and e[1] <= t1]
pts = []
for s, e, n, st in inside:
    heavy = ('blend_fwd' in n) or ('blend_bwd' in n)
    pts.append((s, 1, heavy)); pts.append((e, -1, heavy))
pts.sort()
nall = nheavy = 0
last = t0
acc = {}
for t, d, h in pts:
    key = ("%d blend" % nheavy) + (" + other" if nall > nheavy else "") if nall else "idle"
    acc[key] = acc.get(key, 0) + (t - last)
    last = t
    nall += d
    if h: nheavy += d
print("step %.3f ms" % ((t1 - t0) / 1e6))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print("  %-18s %.3f ms" % (k, v / 1e6))
for st in sorted(set(e[3] for e in inside)):
    es = [e for e in inside if e[3] == st]
    print("  stream %s: %3d kernels, busy %.3f ms, first %.3f, last %.3f" % (st, len(es), sum(e[1] - e[0] for e in es) / 1e6, (es[0][0] - t0) / 1e6, (es[-1][1] - t0) / 1e6))
if len(sys.argv) > 2:
    for s, e, n, st in inside:
        print("%8.3f %8.3f s%s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, st, n.split('(')[1 if n.startswith('(anon') else 0][:40] if False else n[:60]))
