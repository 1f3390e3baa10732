"""Offline fit of the GEMM tile/split cost model (gemm.hip) against measured per-tile timings (gemm_bench logs)."""
import itertools, math, re, sys
sys.path.insert(0, 'scripts')
from gemm_bench import SHAPES

def parse(path):
    out, tile = {}, None
    for l in open(path):
        if l.startswith('=='):
            tile = int(l.split('=')[-1])
        elif 'TF/s' in l:
            name = l.split(' al=')[0].strip()
            us = float(l.split()[-4])
            out.setdefault(name, {})[tile] = us
    return out

CAND = [(128, 128), (128, 64), (64, 64), (128, 96), (96, 128)]
RESID = [3, 4, 7, 4, 4]

def choose(M, N, K, al, bl, P, force=None):
    keq, rate, Lm = P['keq'], P['rate'], P['Lm']
    ktiles = -(-K // 32)
    out = M * N
    max_split = int(0.5 * (M + N) * K / out)
    max_split = max(1, min(max_split, ktiles // 2, 512, (64 << 20) // (out * 4)))
    best = None
    for c, (bm, bn) in enumerate(CAND):
        if force is not None and c != force: continue
        t = -(-M // bm) * -(-N // bn)
        sp_fill = 1
        if t < 256: sp_fill = min(-(-512 // t), max_split)
        sp_fill = max(1, min(sp_fill, max(ktiles, 1)))
        sps = [1, sp_fill] + ([2, 4] if t >= 256 else [])
        for sp in sps:
            sp = max(1, min(sp, max_split, max(ktiles, 1)))
            blocks = t * sp
            avg = blocks / 256.0
            R = RESID[c]
            waves = math.ceil(avg / R) if avg > R else 1
            conc = min(max(avg, 1.0), R) if avg <= R else avg / waves
            nkt = -(-max(ktiles, 1) // sp)
            tm = bm * bn * 32.0 / rate[c]
            per_kt = max(conc * tm, Lm)
            cost = waves * (nkt * per_kt + conc * bm * bn * keq[c] / rate[c])
            if sp > 1: cost += 0.4 * sp * out + 1.2e6
            if best is None or cost < best[0]: best = (cost, c, sp)
    return best

def evaluate(P, meas, verbose=False):
    regret, tot = 0.0, 0.0
    for al, bl, M, N, K, name in SHAPES:
        if name not in meas or len(meas[name]) < len(CAND) or M <= 64: continue     # <= 64 rows: skinny kernel
        cost, c, sp = choose(M, N, K, al, bl, P)
        t_best = min(meas[name].values())
        t_sel = meas[name][c + 1]
        regret += t_sel - t_best; tot += t_best
        if verbose: print(f'{name:28s} pick {CAND[c]} sp={sp:3d}  {t_sel:7.0f}us  best {t_best:7.0f}us {"" if t_sel==t_best else "  <-- +%.0f%%" % (100*(t_sel/t_best-1))}')
    return regret, tot

if __name__ == '__main__':
    meas = parse(sys.argv[1])
    # tiles 0-2 keep the parameters fitted on the 3-tile data (r01_gemm_shapes.txt); the two 96-wide tiles are searched
    base = dict(keq=[150, 60, 45, 45, 45], rate=[1.0, 0.9, 0.8, 0.75, 0.75], Lm=1000e3)
    print('current regret', evaluate(base, meas))
    best = (1e18, None)
    grid = [45, 60, 100, 150, 200, 300], [0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9]
    for k3, k4, r3, r4 in itertools.product(grid[0], grid[0], grid[1], grid[1]):
        P = dict(keq=[150, 60, 45, k3, k4], rate=[1.0, 0.9, 0.8, r3, r4], Lm=1000e3)
        r, t = evaluate(P, meas)
        if r < best[0]: best = (r, P)
    print('best', best)
    evaluate(best[1], meas, verbose=True)
