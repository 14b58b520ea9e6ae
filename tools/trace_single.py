"""Timeline of blocking single-frame calls from a rocprofv3 --kernel-trace CSV of tools/latency_probe.py: per call (a run of kernels
that starts with k_bgr2gray) the kernels' start offsets, durations and the gaps between them; medians over the calls."""
import csv, sys, re, collections
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows if 'at::' not in r['Kernel_Name'])
def short(n):
    m = re.search(r'k_\w+', n)
    return m.group(0) if m else n[:24]
calls, cur = [], []
for s, e, n in ks:
    if 'k_bgr2gray' in n and cur:
        calls.append(cur); cur = []
    cur.append((s, e, short(n)))
calls = [c for c in calls[len(calls) // 2:] if len(c) == len(calls[-1])]
print(len(calls), 'calls of', len(calls[-1]), 'kernels')
n = len(calls[-1])
for i in range(n):
    off = np.median([c[i][0] - c[0][0] for c in calls]) / 1e3
    dur = np.median([c[i][1] - c[i][0] for c in calls]) / 1e3
    gap = np.median([c[i][0] - c[i - 1][1] for c in calls]) / 1e3 if i else 0.0
    print('%-24s start %6.1f us  runs %5.1f us  gap before %5.1f us' % (calls[-1][i][2], off, dur, gap))
print('first start -> last end: %.1f us; sum of kernels %.1f us' % (np.median([c[-1][1] - c[0][0] for c in calls]) / 1e3, np.median([sum(e - s for s, e, _ in c) for c in calls]) / 1e3))
