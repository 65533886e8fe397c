"""HIP bf16 step vs its torch restatement (tests/sim_lib.py) per residual-stream form: per-tensor gradient distance, largest first."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np, torch
import sim_lib
import adapter4rec_amd.engine as E
from test_engine_gpu import build, condition, strip

for rd in ('bf16', 'bf24', 'bf20'):
    root, args, sd, cfg, fx, items, mask = build('houlsby', 'bf16')
    inner = getattr(root, 'model', root)
    inner.args.residual_dtype = rd
    inner.invalidate_native()
    sd = condition(sd)
    full = {str(k): sd[strip(str(k))] for k in fx['all_keys']}
    root.load_state_dict(full, strict=True)
    loss = root(items, mask, 0); loss.backward()
    g_gpu = {n: p.grad.cpu().clone() for n, p in root.named_parameters() if p.requires_grad}
    l_gpu = loss.item()
    real_L, real_req = E.L, E.TransRecEngine._require_device
    try:
        E.L = sim_lib
        E.TransRecEngine._require_device = lambda self, p0: None
        root.cpu(); root.load_state_dict(full, strict=True)
        for p in root.parameters(): p.grad = None
        loss_c = root(items.cpu(), mask.cpu(), 'cpu'); loss_c.backward()
    finally:
        E.L, E.TransRecEngine._require_device = real_L, real_req
    errs = sorted(((float(np.abs(g_gpu[n].numpy() - p.grad.numpy()).max() / (np.abs(p.grad.numpy()).max() + 1e-12)), n) for n, p in root.named_parameters() if p.requires_grad), reverse=True)
    print(rd, 'loss', l_gpu, loss_c.item(), [(round(e, 4), n[-60:]) for e, n in errs[:4]])
