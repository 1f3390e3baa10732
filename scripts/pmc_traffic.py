"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

    python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps> <csrc_sha> > profiles/rNN_pmc_traffic.json

Units / corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KB; on gfx950 FETCH_SIZE
reports half of the bytes of wide (16 B/lane) coalesced reads, so it is doubled for the GEMM kernels (whose operand
loads are all 16-byte); WRITE_SIZE is taken as is (uncalibrated).  Values are means per launch."""
import collections
import csv
import json
import re
import sys


STEADY = {}      # counter -> (KB inside the steady-state steps, number of such steps)
DUR = collections.defaultdict(lambda: [0, 0])      # kernel -> [ns, dispatches] in the FETCH_SIZE pass (serialised dispatches: undisturbed durations)


def load(path, counter):
    """Per-kernel sums over ALL dispatches, and - for the per-step total - the bytes of the dispatches between the end of
    the first gradient step and the end of the last one (4 adamw_kernel launches close a step): the benchmark's own
    set-up kernels (synthetic replay ring, parameter initialisation) and the first step's lazy allocations stay out."""
    acc = collections.defaultdict(lambda: [0.0, 0])
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    for r in rows:
        name = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('(anonymous namespace)::', '')).replace('void ', '')
        a = acc[name]
        a[0] += float(r['Counter_Value'])
        a[1] += 1
        if counter == 'FETCH_SIZE' and r.get('Start_Timestamp') and r.get('End_Timestamp'):      # dispatches run one at a time under PMC
            d = DUR[name]
            d[0] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            d[1] += 1
    if rows and 'Dispatch_Id' in rows[0]:
        rows.sort(key=lambda r: int(r['Dispatch_Id']))
        ad = [int(r['Dispatch_Id']) for r in rows if r['Kernel_Name'].startswith('adamw_kernel')]
        if len(ad) >= 8 and len(ad) % 4 == 0:
            lo, hi = ad[3], ad[-1]
            STEADY[counter] = (sum(float(r['Counter_Value']) for r in rows if lo < int(r['Dispatch_Id']) <= hi), len(ad) // 4 - 1)
    return acc


fetch = load(sys.argv[1], 'FETCH_SIZE')
write = load(sys.argv[2], 'WRITE_SIZE')
out = {}
for name in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(name, [0.0, 0])
    w, nw = write.get(name, [0.0, 0])
    fb = 1024.0 * f / max(nf, 1)
    wb = 1024.0 * w / max(nw, 1)
    out[name] = dict(launches=max(nf, nw), fetch_bytes_per_launch_raw=fb, fetch_bytes_per_launch=2.0 * fb,
                     write_bytes_per_launch=wb, hbm_bytes_per_launch=2.0 * fb + wb)
    if DUR[name][1]:      # mean duration of a launch (counter pass: dispatches serialised) and the HBM rate that goes with it
        us = DUR[name][0] / DUR[name][1] / 1e3
        out[name].update(avg_us=us, hbm_gb_per_s=(2.0 * fb + wb) / (us * 1e-6) / 1e9 if us > 0 else None)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3        # bench.py --steps S --warmup W --prof-steps 0: S + W steps in total
sha = sys.argv[4] if len(sys.argv) > 4 else None            # `python bench.py --csrc-sha`: fingerprint of the kernel sources
total = sum(v['hbm_bytes_per_launch'] * v['launches'] for v in out.values()) / steps / 1e9
total_all = total
scope = 'all dispatches of the process / steps (includes the benchmark set-up kernels)'
if 'FETCH_SIZE' in STEADY and 'WRITE_SIZE' in STEADY and STEADY['FETCH_SIZE'][1] == STEADY['WRITE_SIZE'][1] >= 1:
    total = 1024.0 * (2.0 * STEADY['FETCH_SIZE'][0] + STEADY['WRITE_SIZE'][0]) / STEADY['FETCH_SIZE'][1] / 1e9
    scope = f"dispatches after the first gradient step only ({STEADY['FETCH_SIZE'][1]} step(s)); with set-up kernels: {total_all:.2f}"
steps_total = STEADY['FETCH_SIZE'][1] + 1 if 'FETCH_SIZE' in STEADY else steps      # gradient steps the process ran (4 adamw launches close one)
json.dump(dict(note='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), bench.py --steps 2 '
                    '--warmup 1 --no-overlap --prof-steps 0; FETCH_SIZE x2 (gfx950 wide-read correction), KB -> bytes',
               csrc_sha=sha, steps_profiled=steps, steps_total=steps_total, total_gb_per_step=round(total, 2), total_scope=scope, kernels=out),
          sys.stdout, indent=1)
