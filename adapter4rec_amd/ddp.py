"""FlatDDP: the data-parallel wrapper for the native path (replaces torch DDP at run.py:503).

One process per GPU over RCCL (torch.distributed backend 'nccl' on ROCm).  Users are sharded over
ranks by the sampler; the only exchange is ONE all-reduce of the flat fp32 adapter-gradient buffer
per step (9.55 MB for BERT-base + Houlsby), averaged over ranks exactly like DDP.  The frozen
backbone is broadcast once at wrap time and never communicated again.  Exposes ``.module`` because
the reference's eval code dereferences ``model.module.*`` (data_utils/metrics.py:72-75,101-104).
"""
import torch
import torch.distributed as dist
from torch import nn


def any_rank(flag, group=None):
    """Logical OR of a host or 0-d device flag over the ranks of `group` (one MAX all-reduce); the flag itself without a process group."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return bool(flag)
    dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
    t = torch.as_tensor(flag).to(dev).reshape(1).to(torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(t.item())


class FlatDDP(nn.Module):
    def __init__(self, module, device_ids=None, output_device=None, find_unused_parameters=False, process_group=None,
                 broadcast=True):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == 'nccl'
        self._pending = []
        self._order, self.last_order_hash = 0, 0              # rolling hash of this step's launch_ ranges / of the last completed step's
        if self.world > 1 and broadcast:                       # DDP constructor semantics: rank 0's state everywhere
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, 0, group=process_group)
            inner = getattr(module, 'model', module)
            if hasattr(inner, 'invalidate_native'):
                inner.invalidate_native()
        inner = getattr(module, 'model', module)               # CompacterModel keeps the TransRec model in .model
        # plain attribute, NOT a registered sub-module: nn.Module.__setattr__ would make module <-> wrapper a cycle and
        # .train() / .eval() / .named_parameters() recurse forever on world > 1
        # (A4R_DDP_FORCE=1: run the exchange with a single rank too -- exercises the RCCL calls on a one-GPU box)
        force = bool(int(__import__('os').environ.get('A4R_DDP_FORCE', '0'))) and dist.is_initialized()
        object.__setattr__(inner, '_a4r_ddp', self if (self.world > 1 or force) else None)

    def average_(self, flat):
        """In-place mean over ranks of one flat gradient buffer (a single RCCL all-reduce)."""
        if self._avg:                                          # RCCL averages inside the collective: one launch
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.process_group)
        else:                                                  # gloo (CPU tests) has no AVG
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.process_group)
            flat.mul_(1.0 / self.world)
        return flat

    # ---- chunked exchange, overlapped with backward (the reference's DDP buckets, run.py:503,599)
    def launch_(self, flat, lo, hi):
        """Start the mean over ranks of flat[lo:hi] (gradients that are final) without blocking the launching stream: the
        collective runs on the process group's own stream behind everything already enqueued; wait_all() joins it.  Every rank
        launches the same ranges in the same order (the engine derives them from the parameter list)."""
        if hi <= lo:
            return
        # launch-order fingerprint of the step (every rank must launch the same ranges in the same order: bench.py gathers it)
        self._order = (self._order * 1000003 + (lo * 2654435761 + hi)) & 0x7FFFFFFFFFFFFFFF
        view = flat[lo:hi]
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self._pending.append((dist.all_reduce(view, op=op, group=self.process_group, async_op=True), view))

    def wait_all(self):
        """Join every chunk launched since the last call.  Always runs (engine.backward_bound calls it from a `finally`): a backward
        that raised after some launches must not leave works in flight whose 1/world scaling (gloo) would be applied to a later step."""
        pending, self._pending = self._pending, []
        self.last_order_hash, self._order = self._order, 0
        for work, view in pending:
            work.wait()
            if not self._avg:
                view.mul_(1.0 / self.world)

    def any_rank(self, flag):
        """True on EVERY rank when `flag` is true on any of them (the end-of-epoch NaN test of run.py:601-603 decides whether the ranks
        enter the evaluation's collectives: a rank-local decision would leave the others waiting in an all-gather)."""
        return any_rank(flag, self.process_group)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
