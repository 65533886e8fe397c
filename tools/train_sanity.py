#!/usr/bin/env python
"""300 steps of the bench workload on a FIXED small user set: the training loss must fall (bf16, dropout on, FusedAdam, 4 lr groups)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else 'bert_houlsby'
dev = torch.device('cuda:0')
if wl == 'vit_lora':
    args = bench.make_cv_args(4, 'bf16', wl)
    model, opt = bench.build_cv_model(args, dev)
    batches = bench.synth_image_batches(4, 2, dev, 1)
    eng = getattr(model, 'model', model)._engine()
else:
    args = bench.make_args(16, 'bf16')
    args.lr, args.adapter_bert_lr, args.adapter_sasrec_lr = 1e-3, 1e-3, 1e-3
    model, opt = bench.build_model(args, dev)
    g = torch.Generator().manual_seed(1); gc = torch.Generator().manual_seed(2)
    content = bench.synth_content(4096, gc)
    batches = [(i.to(dev), m.to(dev)) for i, m in bench.synth_batches(content, 4096, 16, 2, g)]
    eng = model._engine()
hist = []
for step in range(300):
    items, mask = batches[step % len(batches)]
    eng.flat_g.zero_()
    loss = eng.train_forward(items, mask)
    eng.train_backward(into_flat_grad=True)
    opt.step()
    if step % 50 == 0 or step == 299:
        hist.append(float(loss))
        print(f'step {step}: loss {float(loss):.4f}', flush=True)
assert hist[-1] < hist[0] - 0.05, hist
print('loss falls:', hist)
