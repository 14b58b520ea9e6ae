"""How kernel A fares beside kernel B in a rocprofv3 --kernel-trace CSV: duration of A's launches by the share of them that ran while a B launch was running.
usage: trace_overlap.py <kernel_trace.csv> <substring of A> <substring of B>"""
import csv, sys, bisect
rows = list(csv.DictReader(open(sys.argv[1])))
A = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if sys.argv[2] in r['Kernel_Name']]
B = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if sys.argv[3] in r['Kernel_Name'])
A = A[len(A) // 3:]
bins = {}
for s, e in A:
    ov = 0
    for bs, be in B:
        if be <= s: continue
        if bs >= e: break
        ov += min(e, be) - max(s, bs)
    f = min(1.0, ov / max(1, e - s))
    k = 0 if f < 0.1 else (1 if f < 0.9 else 2)
    bins.setdefault(k, []).append((e - s) / 1e3)
for k, name in ((0, 'alone (< 10 % overlapped)'), (1, 'partly'), (2, 'inside B (> 90 %)')):
    v = bins.get(k, [])
    if v: print('%-28s %5d launches, mean %.1f us, median %.1f us' % (name, len(v), sum(v) / len(v), sorted(v)[len(v) // 2]))
