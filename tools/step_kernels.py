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
if wl in ('vit_lora', 'mae_compacter'):
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
