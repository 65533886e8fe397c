#!/usr/bin/env python
"""Where do the step's launches that are NOT liba4r_hip.so kernels come from?  One steady-state step under torch.profiler with Python
stacks: (1) every device activity up to the first 256-tile GEMM (the step's prelude) with offset + duration, (2) the aten ops that put a
kernel / memcpy / memset on the device, grouped by the engine.py / ddp.py / optim.py line that issued them.
usage: python tools/step_small_ops.py [workload] [dtype]"""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_argv = sys.argv
sys.argv = ['x'] + _argv[1:]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'step_kernels.py')).read().split("for i in range(4):")[0])
for i in range(4):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
    step(4)
    torch.cuda.synchronize()
evs = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
names = [e.name for e in evs]
first = min(i for i, n in enumerate(names) if 'gemm_nt_256' in n)
t0 = evs[0].time_range.start
print(f'{len(evs)} device activities; prelude = {first} activities, {evs[first].time_range.start - t0:.0f} us until the first 256-tile GEMM')
for e in evs[:first]:
    print(f'  {e.time_range.start - t0:8.1f} +{e.time_range.end - e.time_range.start:6.1f}  {e.name[:110]}')

ours = lambda n: ('at::native' not in n and 'rocclr' not in n and 'Cijk' not in n and 'Memset' not in n and 'Memcpy' not in n)
by_site = defaultdict(lambda: [0, 0.0, set()])
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    foreign = [k for k in e.kernels if not ours(k.name)]
    if not foreign or not e.name.startswith('aten::'):
        continue
    site = next((s for s in (e.stack or []) if 'adapter4rec_amd' in s or 'bench.py' in s), '?')
    rec = by_site[(site.strip()[:120], e.name)]
    rec[0] += len(foreign)
    rec[1] += sum(k.duration for k in foreign)
    rec[2].update(k.name[:50] for k in foreign)
print('\nnon-a4r device work by issuing line (launches, device us, op):')
for (site, op), (n, us, ks) in sorted(by_site.items(), key=lambda kv: -kv[1][1]):
    print(f'  {n:4d} {us:8.1f}  {op:28s} {site}   {sorted(ks)[:2]}')
