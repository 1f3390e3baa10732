"""Per-millisecond timeline + per-kernel totals of ONE gradient step from a rocprofv3 --kernel-trace csv.
usage: python scripts/trace_timeline.py <kernel_trace.csv> [step_from_end=2]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
ad = [e for e in ev if 'adamw' in e[2]]
bounds = [ad[i][1] for i in range(3, len(ad), 4)]
s, e = bounds[-back - 1], bounds[-back]
st = [x for x in ev if x[0] >= s and x[1] <= e + 1000]
busy, cs, ce = 0, None, None
for a, b, _, _ in st:
    if ce is None or a > ce:
        if ce is not None:
            busy += ce - cs
        cs, ce = a, b
    else:
        ce = max(ce, b)
busy += ce - cs
print(f'step {1e-6 * (e - s):.2f} ms, {len(st)} kernels, busy union {1e-6 * busy:.2f} ms, sum {1e-6 * sum(b - a for a, b, _, _ in st):.2f} ms')
tot = collections.Counter()
cnt = collections.Counter()
durs = collections.defaultdict(list)
for a, b, n, q in st:
    tot[n[:70]] += b - a
    cnt[n[:70]] += 1
    durs[n[:70]].append(b - a)
for n, d in tot.most_common(22):
    ds = sorted(durs[n])
    print(f'  {n:70s} {cnt[n]:5d} {d / 1e6:7.2f} ms {d / cnt[n] / 1e3:8.1f} us  (median {ds[len(ds) // 2] / 1e3:.1f}, min {ds[0] / 1e3:.1f})')
bins = collections.defaultdict(collections.Counter)
for a, b, n, q in st:
    bins[int((a - s) / 1e6)][(n[:34], q)] += b - a
for k in sorted(bins):
    print(k, ' | '.join(f'{n} q{q} {d / 1e3:.0f}us' for (n, q), d in bins[k].most_common(3)))
