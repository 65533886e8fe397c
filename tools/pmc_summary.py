#!/usr/bin/env python
"""Per-kernel fabric/HBM traffic from two rocprofv3 PMC passes over the same command (one counter per pass, --kernel-trace only):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir_fetch> -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d <dir_write> -o w --output-format csv -- python3 bench.py ... (same)
    python tools/pmc_summary.py <dir_fetch> <dir_write> > profiles/<round>_pmc_hbm_traffic.json

Counters are KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-B requests of wide
coalesced reads at 64 B, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact.  Infinity-Cache hits are counted: this is fabric
traffic, an upper bound on HBM traffic."""
import csv, glob, json, os, sys, collections

def load(d, counter):
    f = sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True))
    assert f, f'no counter_collection.csv under {d}'
    acc = collections.defaultdict(lambda: [0, 0.0])
    for path in f:
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] != counter:
                continue
            a = acc[r['Kernel_Name']]
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    return acc

fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
out = {'how': __doc__.strip(), 'kernels': {}}
for k in sorted(fetch, key=lambda k: -fetch[k][1]):
    if k not in write or fetch[k][0] == 0:
        continue
    n = fetch[k][0]
    rd = 2.0 * fetch[k][1] * 1024 / n
    wr = write[k][1] * 1024 / max(write[k][0], 1)
    if rd + wr < 1e6:
        continue
    out['kernels'][k] = dict(launches=n, read_bytes_per_launch=int(rd), write_bytes_per_launch=int(wr), traffic_bytes_per_launch=int(rd + wr))
json.dump(out, sys.stdout, indent=1)
