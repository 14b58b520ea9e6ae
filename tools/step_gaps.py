"""GPU idle time inside the steps of a bench run: python tools/step_gaps.py <rocprofv3 -d dir> <kernel substring that marks a step's work>
Splits the kernel trace into steps at idle periods > 5 ms and prints, per step, wall span, union-busy time and the largest gaps."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
key = sys.argv[2] if len(sys.argv) > 2 else ''
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:24])
            for r in csv.DictReader(open(f)))
steps, cur, end = [], [], None
for e in ev:
    if end is not None and e[0] - end > 5e6:
        steps.append(cur); cur = []
    cur.append(e); end = e[1] if end is None else max(end, e[1])
steps.append(cur)
for si, st in enumerate(steps):
    if key and not any(key in e[2] for e in st):
        continue
    t0, t1 = st[0][0], max(e[1] for e in st)
    busy, cs, ce, gaps = 0, None, None, []
    for s, e, n in st:
        if ce is None or s > ce:
            if ce is not None:
                busy += ce - cs
                gaps.append(((s - ce) / 1e3, (ce - t0) / 1e6, n))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    gaps.sort(reverse=True)
    print('step %d: %d kernels, span %.2f ms, busy %.2f ms, idle %.2f ms; largest gaps (us @ ms, next kernel): %s' % (
        si, len(st), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, [(round(g, 0), round(a, 1), n) for g, a, n in gaps[:6]]))
