"""Per-queue view of ONE gradient step from a rocprofv3 --kernel-trace csv: which stream is busy when, and with what.
usage: python scripts/trace_queues.py <kernel_trace.csv> [step_from_end=2] [bin_ms=1.0]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
binms = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
ad = [e for e in ev if 'adamw' in e[2]]
bounds = [ad[i][1] for i in range(3, len(ad), 4)]
s, e = bounds[-back - 1], bounds[-back]
st = [x for x in ev if x[0] >= s and x[1] <= e + 1000]
queues = sorted({q for _, _, _, q in st})
print(f'step {1e-6 * (e - s):.2f} ms, {len(st)} kernels, queues {queues}')


def short(n):
    n = n.replace('void ', '')
    return n[:n.index('(')] if '(' in n and n.index('(') < 44 else n[:44]


for q in queues:
    ks = [x for x in st if x[3] == q]
    busy = sum(b - a for a, b, _, _ in ks)
    print(f'queue {q}: {len(ks)} kernels, busy {busy / 1e6:.2f} ms, first {1e-6 * (ks[0][0] - s):.2f} ms, last end {1e-6 * (ks[-1][1] - s):.2f} ms')
nb = int((e - s) / 1e6 / binms) + 1
print('bin  ' + ' | '.join(f'q{q}: busy%  top kernel' for q in queues))
for b in range(nb):
    lo, hi = s + b * binms * 1e6, s + (b + 1) * binms * 1e6
    cells = []
    for q in queues:
        tot = collections.Counter()
        for a, c, n, qq in st:
            if qq == q and c > lo and a < hi:
                tot[short(n)] += min(c, hi) - max(a, lo)
        if tot:
            n, d = tot.most_common(1)[0]
            cells.append(f'{100 * sum(tot.values()) / (binms * 1e6):4.0f}% {n[:34]:34s}')
        else:
            cells.append(' ' * 40)
    print(f'{b * binms:5.1f} ' + ' | '.join(cells))
# ---- every idle gap > 0.3 ms inside a queue's active span: what ended last on the OTHER queues before the gap closed
print('gaps > 0.3 ms inside a queue (start of the kernel that ends the gap; last kernels to end on the other queues before it):')
for q in queues:
    ks = [x for x in st if x[3] == q]
    for p, n in zip(ks, ks[1:]):
        if n[0] - p[1] > 300000:
            others = []
            for qq in queues:
                if qq == q:
                    continue
                prev = [x for x in st if x[3] == qq and x[1] <= n[0]]
                if prev:
                    o = max(prev, key=lambda x: x[1])
                    others.append(f'q{qq} {short(o[2])[:30]} ended {1e-6 * (o[1] - s):.2f}')
            print(f'  q{q}: idle {1e-6 * (p[1] - s):.2f} -> {1e-6 * (n[0] - s):.2f} ms, then {short(n[2])[:40]} (after {short(p[2])[:30]}); ' + '; '.join(others))
for q in queues:      # ... and what ended last elsewhere before each queue's FIRST kernel of the step
    k0 = [x for x in st if x[3] == q][0]
    others = []
    for qq in queues:
        prev = [x for x in st if x[3] == qq and x[1] <= k0[0]]
        if qq != q and prev:
            o = max(prev, key=lambda x: x[1])
            others.append(f'q{qq} {short(o[2])[:30]} ended {1e-6 * (o[1] - s):.3f}')
    print(f'  q{q} starts {1e-6 * (k0[0] - s):.3f} with {short(k0[2])[:30]}; ' + '; '.join(others))
for q in queues:
    ks = [x for x in st if x[3] == q][:4]
    print(f'  q{q} first kernels: ' + ', '.join(f'{short(x[2])[:28]}@{1e-6 * (x[0] - s):.2f}' for x in ks))
