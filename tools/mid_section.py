#!/usr/bin/env python
"""The part of one training step between the item tower's forward and its backward (item head, SASRec user tower forward, loss, SASRec
backward, item head backward): every device activity with its start offset and duration (torch.profiler), and what follows the last large
kernel (last weight gradients, Adam).  ~70 launches of 5 - 25 us each: pure launch latency.   usage: python tools/mid_section.py"""
import os, sys
sys.argv = ['x']
exec(open('tools/step_kernels.py').read().split("evs = sorted(")[0])
evs = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
names = [e.name for e in evs]
i0 = max(i for i, n in enumerate(names) if 'adapter_ln_fwd' in n)
i1 = min(i for i, n in enumerate(names) if 'adapter_ln_bwd' in n)
span = evs[i1].time_range.start - evs[i0].time_range.end
busy = sum(e.time_range.end - e.time_range.start for e in evs[i0 + 1:i1])
print(f'between the last adapter_ln_fwd and the first adapter_ln_bwd: {i1 - i0 - 1} device activities, span {span:.0f} us, busy {busy:.0f} us')
for e in evs[i0 + 1:i1]:
    print(f'  {e.time_range.start - evs[i0].time_range.end:8.1f} +{e.time_range.end - e.time_range.start:6.1f}  {e.name[:100]}')
t0 = evs[0].time_range.start
print('first kernel', names[0][:60], 'until first gemm_nt_256:', evs[min(i for i,n in enumerate(names) if 'gemm_nt_256' in n)].time_range.start - t0)
last256 = max(i for i, n in enumerate(names) if 'gemm_nt_256' in n or 'adapter_ln_bwd' in n)
print('after the last large kernel:', len(evs) - 1 - last256, 'activities,', evs[-1].time_range.end - evs[last256].time_range.end, 'us')
for e in evs[last256 + 1:]:
    print(f'  +{e.time_range.end - e.time_range.start:6.1f}  {e.name[:100]}')
