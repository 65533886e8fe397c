#!/usr/bin/env python
"""Wall time of the user-encoder (SASRec) sections inside one training step: launch-bound tiny kernels?  (bench workload, B = 32)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
args = bench.make_args(32, 'bf16')
model, opt = bench.build_model(args, dev)
g = torch.Generator().manual_seed(1); gc = torch.Generator().manual_seed(2)
content = bench.synth_content(65536, gc)
batches = [(i.to(dev), m.to(dev)) for i, m in bench.synth_batches(content, 65536, 32, 2, g)]
eng = model._engine()
rec = {}
def timed(name, fn):
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        rec.setdefault(name, []).append((e0, e1))
        return r
    return w
eng._user_forward = timed('user_forward', eng._user_forward)
eng._items_backward = timed('items_backward', eng._items_backward)
eng._encode = timed('encode_items', eng._encode)
def step(i):
    items, mask = batches[i % 2]
    eng.flat_g.zero_()
    loss = eng.train_forward(items, mask)
    eng.train_backward(into_flat_grad=True)
    opt.step()
for i in range(5): step(i)
rec.clear()
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for i in range(10): step(i)
t1.record(); torch.cuda.synchronize()
print('step', t0.elapsed_time(t1) / 10, 'ms')
for k, v in rec.items():
    print(k, sum(a.elapsed_time(b) for a, b in v) / 10, 'ms per step')
