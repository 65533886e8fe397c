#!/usr/bin/env python
"""Matrix-pipe utilisation per kernel from rocprofv3 PMC passes over the bench step (one counter per pass, --kernel-trace only):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d <dir_m> -o m --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc GRBM_GUI_ACTIVE          --kernel-trace -d <dir_g> -o g --output-format csv -- python3 bench.py ... (same)
    python tools/pmc_mfma.py <dir_m> <dir_g> > profiles/<round>_pmc_mfma_util.json

SQ_VALU_MFMA_BUSY_CYCLES sums, over the chip's 1 024 SIMDs, the cycles the matrix pipe is busy; GRBM_GUI_ACTIVE sums the active cycles
of the 8 XCDs.  utilisation = (MFMA_BUSY / 1024) / (GUI_ACTIVE / 8): the fraction of a kernel's cycles in which a SIMD's matrix pipe
holds an MFMA (what /opt/skills/guides/MI355X_MICROARCH.md calls MFMA utilisation; 1.0 = the dense peak at the clock the kernel ran at)."""
import collections, csv, glob, json, os, sys


def load(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for path in sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)):
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] == counter:
                a = acc[r['Kernel_Name']]
                a[0] += 1
                a[1] += float(r['Counter_Value'])
    return acc


m, g = load(sys.argv[1], 'SQ_VALU_MFMA_BUSY_CYCLES'), load(sys.argv[2], 'GRBM_GUI_ACTIVE')
out = {'how': __doc__.strip(), 'kernels': {}}
for k in sorted(m, key=lambda k: -m[k][1]):
    if k not in g or m[k][1] <= 0:
        continue
    busy, act = m[k][1] / m[k][0] / 1024.0, g[k][1] / g[k][0] / 8.0
    out['kernels'][k] = dict(launches=m[k][0], mfma_busy_cycles_per_simd=round(busy), active_cycles=round(act), mfma_utilisation=round(busy / act, 4))
json.dump(out, sys.stdout, indent=1)
