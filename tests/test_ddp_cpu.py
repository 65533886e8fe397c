"""CPU, world_size 2 over gloo: the data-parallel wrapper averages the flat adapter-gradient buffer exactly like DDP
(mean over ranks of per-rank mean losses) and keeps the replicas in lock-step through FusedAdam.
Kernels are simulated (tests/sim_lib.py); this covers the N > 1 host path that bench.py --gpus N runs over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sim_lib
    import adapter4rec_amd.engine as E
    import adapter4rec_amd.optim as O
    E.L = sim_lib
    O.L = sim_lib
    E.TransRecEngine._require_device = lambda self, p0: None
    from test_engine_host_logic import build_cpu
    from adapter4rec_amd.ddp import FlatDDP
    from adapter4rec_amd.inject import optimizer_groups
    from adapter4rec_amd.optim import FusedAdam
    torch.manual_seed(100 + rank)                       # ranks start from DIFFERENT adapter values: the wrapper must broadcast rank 0's
    root, args, fx, items, mask = build_cpu('houlsby')
    with torch.no_grad():
        for p in root.parameters():
            if p.requires_grad:
                p.add_(0.01 * rank)
    model = FlatDDP(root)
    opt = FusedAdam(optimizer_groups(model, args))
    B = items.shape[0] // 42
    half = B // 2
    my_items = items.view(B, 42, 60)[rank * half:(rank + 1) * half].reshape(-1, 60)      # users sharded over ranks
    my_mask = mask[rank * half:(rank + 1) * half]
    opt.zero_grad()
    loss = model(my_items, my_mask, 'cpu')
    loss.backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.requires_grad}
    opt.step()
    params = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    torch.save(dict(loss=loss.item(), grads=grads, params=params), os.path.join(out_dir, f'r{rank}.pt'))
    dist.destroy_process_group()


def test_two_rank_gradient_average(tmp_path):
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / 'r0.pt')
    r1 = torch.load(tmp_path / 'r1.pt')
    for k in r0['grads']:
        torch.testing.assert_close(r0['grads'][k], r1['grads'][k], rtol=0, atol=0)       # identical averaged gradients
        torch.testing.assert_close(r0['params'][k], r1['params'][k], rtol=0, atol=0)     # replicas stay in lock-step
    # oracle for the reduction: mean over ranks of each rank's own-gradient (computed single-process)
    sys.path.insert(0, HERE)
    import sim_lib
    import adapter4rec_amd.engine as E
    real_L, real_req = E.L, E.TransRecEngine._require_device
    try:
        E.L = sim_lib
        E.TransRecEngine._require_device = lambda self, p0: None
        from test_engine_host_logic import build_cpu
        acc = None
        for rank in range(2):
            root, args, fx, items, mask = build_cpu('houlsby')
            B = items.shape[0] // 42
            half = B // 2
            it = items.view(B, 42, 60)[rank * half:(rank + 1) * half].reshape(-1, 60)
            loss = root(it, mask[rank * half:(rank + 1) * half], 'cpu')
            loss.backward()
            g = {'module.' + n: p.grad.clone() for n, p in root.named_parameters() if p.requires_grad}
            acc = g if acc is None else {k: acc[k] + g[k] for k in g}
        for k in acc:
            np.testing.assert_allclose(r0['grads'][k].numpy(), (acc[k] / 2).numpy(), rtol=1e-5, atol=1e-7, err_msg=k)
    finally:
        E.L, E.TransRecEngine._require_device = real_L, real_req
