"""Diagnostic: diff two `bench.py --shape-table` files (A = first, B = second), rows that use the bf16-storage kernel or moved."""
import json, sys
def load(f):
    rows = {}
    for ln in open(f).read().strip().splitlines()[1:]:
        p = ln.split()
        rows[(p[0], p[1], p[2], p[3])] = (p[5], float(p[6]), float(p[7]), float(p[8]))
    return rows
a, b = load(sys.argv[1]), load(sys.argv[2])
ta = tb = 0.0
for k, v in sorted(a.items(), key=lambda kv: -kv[1][2]):
    o = b.get(k)
    if int(v[0]) & 32 or (o and abs(o[2] - v[2]) > 0.03):
        print(k, 'flags', v[0], 'A ms', v[2], 'B ms', o[2] if o else None)
        ta += v[2]; tb += o[2] if o else 0
print('sum A', round(ta, 3), 'sum B', round(tb, 3))
for f in sys.argv[3:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['all_gemm']['ms_per_step'])
