#!/usr/bin/env python
"""GPU idle time between the kernels of a training step, from a rocprofv3 --kernel-trace CSV: the steps are cut at adam_kernel, per step
span = last end - first start, busy = sum of kernel durations (overlap-free on one stream), idle = span - busy; gaps by the kernel that FOLLOWS.
usage: python tools/trace_gaps.py <kernel_trace.csv> [skip_steps=5]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cuts = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2] or 'adam4_kernel' in r[2]]
steps = [(cuts[i] + 1, cuts[i + 1] + 1) for i in range(len(cuts) - 1)][skip:]
tot_span = tot_busy = 0
gap_by = defaultdict(lambda: [0, 0])
pair_by = defaultdict(lambda: [0, 0])                 # (predecessor, successor) of every boundary above 2 us
hist = defaultdict(int)
short = lambda n: n.replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '').split('(')[0].split('<')[0][:44]
for a, b in steps:
    seg = rows[a:b]
    tot_span += seg[-1][1] - seg[0][0]
    tot_busy += sum(e - s for s, e, _ in seg)
    for (s0, e0, n0), (s1, e1, n1) in zip(seg, seg[1:]):
        g = max(0, s1 - e0)
        k = n1.replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '').split('(')[0][:60]
        gap_by[k][0] += g
        gap_by[k][1] += 1
        hist[min(g // 1000, 20)] += 1
        if g > 2000:
            pair_by[(short(n0), short(n1))][0] += g
            pair_by[(short(n0), short(n1))][1] += 1
n = len(steps)
print(f'{n} steps: span {tot_span / n / 1e6:.3f} ms  busy {tot_busy / n / 1e6:.3f} ms  idle {(tot_span - tot_busy) / n / 1e6:.3f} ms ({100 * (tot_span - tot_busy) / tot_span:.1f} %)  '
      f'{sum(c for _, c in gap_by.values()) / n:.0f} launches per step')
print('gap histogram (us: count per step):', {k: round(v / n, 1) for k, v in sorted(hist.items())})
for k, (g, c) in sorted(gap_by.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f'  {g / n / 1e3:8.1f} us per step before {c / n:6.1f} x {k}  (mean {g / c / 1e3:.2f} us)')
print('boundaries above 2 us, by (predecessor -> successor), per step:')
for (a, b), (g, c) in sorted(pair_by.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f'  {g / n / 1e3:8.1f} us in {c / n:5.1f} boundaries (mean {g / c / 1e3:5.2f} us)  {a}  ->  {b}')
