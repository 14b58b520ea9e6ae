"""GPU busy fraction and kernel concurrency of a rocprofv3 --kernel-trace CSV (the timed part: the last two thirds of the dispatches)."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows if 'at::' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name'])
t0 = ks[len(ks) // 3][0]
sel = [k for k in ks if k[0] >= t0]
ev = []
for s, e, _ in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist = collections.Counter(); c = 0; last = ev[0][0]
for t, d in ev:
    hist[c] += t - last; last = t; c += d
tot = sum(hist.values())
print('span ms %.1f; kernels running -> share of time:' % (tot / 1e6), {k: round(v / tot, 3) for k, v in sorted(hist.items())})
dur = collections.Counter()
for s, e, n in sel:
    m = re.search(r'k_\w+(<[^>]*>)?', n)
    dur[m.group(0) if m else n[:30]] += e - s
print('sum of durations / span = %.2f' % (sum(dur.values()) / tot))
for n, v in dur.most_common(8):
    print('  %6.1f %% of span  %s' % (100 * v / tot, n))
