"""What surrounds the all-reduce kernels of a step: from a rocprofv3 --kernel-trace csv of `bench.py --force-dp`, for the
second-to-last step: each RCCL kernel with the kernel before / after it on its queue and what the other queues ran meanwhile;
plus the step's per-kernel totals (compare with the trace of the same command without --force-dp).
usage: python scripts/r06_dp_trace.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
ad = [e for e in ev if 'adamw' in e[2]]
bounds = [ad[i][1] for i in range(3, len(ad), 4)]
s, e = bounds[-3], bounds[-2]
st = [x for x in ev if x[0] >= s and x[1] <= e + 1000]
print(f'step {1e-6 * (e - s):.3f} ms, {len(st)} kernels, sum of durations {1e-6 * sum(b - a for a, b, _, _ in st):.3f} ms')
byq = collections.defaultdict(list)
for x in st:
    byq[x[3]].append(x)
for q, l in sorted(byq.items()):
    print(f'  queue {q}: {len(l)} kernels, busy {1e-6 * sum(b - a for a, b, _, _ in l):.3f} ms, first {1e-6 * (l[0][0] - s):.3f} last end {1e-6 * (l[-1][1] - s):.3f}'
          f'  e.g. ' + ', '.join(n.split('(')[0][-28:] for n, _ in collections.Counter(x[2] for x in l).most_common(3)))
for i, (a, b, n, q) in enumerate(st):
    if 'nccl' not in n.lower() and 'rccl' not in n.lower():
        continue
    l = byq[q]
    j = l.index((a, b, n, q))
    prev = l[j - 1] if j else None
    nxt = l[j + 1] if j + 1 < len(l) else None
    print(f'RCCL kernel at +{1e-6 * (a - s):.3f} ms, {1e-3 * (b - a):.1f} us, queue {q}: {n[:50]}')
    if prev:
        print(f'    before on its queue: {prev[2][:50]} ended {1e-3 * (a - prev[1]):.1f} us earlier')
    if nxt:
        print(f'    after on its queue:  {nxt[2][:50]} started {1e-3 * (nxt[0] - b):.1f} us later')
    lo, hi = (prev[1] if prev else a) , (nxt[0] if nxt else b)
    for q2, l2 in sorted(byq.items()):
        if q2 == q:
            continue
        ov = [(x[2][:30], min(x[1], hi) - max(x[0], lo)) for x in l2 if x[1] > lo and x[0] < hi]
        if ov:
            print(f'    queue {q2} meanwhile ({1e-3 * (hi - lo):.1f} us window): {len(ov)} kernels, {1e-3 * sum(o for _, o in ov):.1f} us busy')
tot, cnt = collections.Counter(), collections.Counter()
for a, b, n, q in st:
    tot[n[:60]] += b - a
    cnt[n[:60]] += 1
for n, d in tot.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 14):
    print(f'  {n:60s} {cnt[n]:5d} {d / 1e6:7.3f} ms')
