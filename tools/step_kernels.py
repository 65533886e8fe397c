#!/usr/bin/env python
"""Which device kernels run inside ONE steady-state training step?  (torch.profiler, after warm-up steps so that one-time buffer
allocations and packing are out of the picture.)  Prints the launches that are not liba4r_hip.so kernels, in launch order with their
position between the first and last a4r kernel.   usage: python tools/step_kernels.py [workload] [dtype]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench

wl = sys.argv[1] if len(sys.argv) > 1 else 'bert_houlsby'
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
B = bench.WORKLOADS[wl][1]
if wl in ('vit_lora', 'mae_compacter', 'mae_pretrain'):
    model, opt = bench.build_cv_model(bench.make_cv_args(B, dtype, wl), dev)
    batches = bench.synth_image_batches(B, 2, dev, 1)
else:
    args = bench.make_args(B, dtype)
    if wl == 'roberta_pfeiffer_cpc':
        args.adapter_type, args.adapter_activation, args.arch, args.bert_model_load = 'pfeiffer', 'relu', 'cpc', 'roberta_base'
    model, opt = bench.build_model(args, dev, roberta=(wl == 'roberta_pfeiffer_cpc'))
    g = torch.Generator().manual_seed(1)
    content = bench.synth_content(65536, g)
    batches = [(i.to(dev), m.to(dev)) for i, m in bench.synth_batches(content, 65536, B, 2, g)]


def step(i):
    items, mask = batches[i % 2]
    opt.zero_grad()
    loss = model(items, mask, 0)
    loss.backward()
    opt.step()


for i in range(4):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    step(4)
    torch.cuda.synchronize()
evs = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
ours = lambda n: ('at::native' not in n and 'rocclr' not in n and 'Cijk' not in n and 'Memset' not in n and 'Memcpy' not in n)
idx = [i for i, e in enumerate(evs) if ours(e.name)]
print(f'{wl} {dtype}: {len(evs)} device activities in one step, {len(idx)} of them liba4r_hip.so kernels')
for i, e in enumerate(evs):
    if not ours(e.name):
        where = 'before first' if i < idx[0] else ('after last' if i > idx[-1] else 'BETWEEN')
        print(f'  [{i:4d}] {where:12s} {e.name[:110]}')

# idle time between consecutive device activities of the busiest stream (kernel-boundary cost: drain + release + dispatch)
by_stream = {}
for e in evs:
    by_stream.setdefault(getattr(e, 'device_index', 0) * 1000 + (getattr(e, 'device_resource_id', None) or 0), []).append(e)
# torch.profiler does not expose the stream of an event portably: fall back to one list and merge overlapping intervals
iv = sorted((e.time_range.start, e.time_range.end) for e in evs)
span = iv[-1][1] - iv[0][0]
busy, cur_s, cur_e, gaps = 0.0, iv[0][0], iv[0][1], []
for s, t in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, t
    else:
        cur_e = max(cur_e, t)
busy += cur_e - cur_s
import numpy as np
gaps = np.array(gaps)
print(f'span first start .. last end {span / 1e3:.2f} ms; device busy (union of all streams) {busy / 1e3:.2f} ms; idle {gaps.sum() / 1e3:.2f} ms in {len(gaps)} gaps '
      f'(median {np.median(gaps):.1f} us, p90 {np.percentile(gaps, 90):.1f} us, max {gaps.max():.1f} us)')
big = sorted(((g, i) for i, g in enumerate(gaps)), reverse=True)[:8]
print('largest gaps (us):', ', '.join(f'{g:.1f}' for g, _ in big))
