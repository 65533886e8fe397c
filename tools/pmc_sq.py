#!/usr/bin/env python
"""Per-kernel averages of whatever counters a set of rocprofv3 --pmc passes collected (one directory per pass):
    python tools/pmc_sq.py <substring of kernel name> <dir> [<dir> ...]
SQ_* cycle counters are quad-cycles summed over waves (MI355X_MICROARCH.md, PMC section); printed per launch."""
import collections, csv, glob, os, sys

pat = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in sys.argv[2:]:
    for path in sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)):
        for r in csv.DictReader(open(path)):
            if pat in r['Kernel_Name']:
                a = acc[r['Kernel_Name'][:70]][r['Counter_Name']]
                a[0] += 1
                a[1] += float(r['Counter_Value'])
for k, cs in acc.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print('   %-36s %16.0f  (%d launches)' % (c, v / n, n))
