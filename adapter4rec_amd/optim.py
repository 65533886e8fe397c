"""FusedAdam: torch.optim.Adam's update (as configured at Downstream/Text/run.py:524-529: betas (0.9, 0.999),
eps 1e-8, no weight decay, per-group lr) as ONE kernel launch over the engine's flat parameter / gradient
buffers instead of ~4 small kernels per tensor.  Same constructor shape as torch.optim.Adam (param groups)."""
import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if weight_decay != 0:
            raise NotImplementedError('weight_decay: the reference uses none (run.py:524-529)')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0))
        self._bound = None
        self._step = 0

    # -- binding to the engine's flat buffers (parameters become views at the engine's first use)
    def _bind(self):
        plist = [p for g in self.param_groups for p in g['params']]
        if not plist:
            raise RuntimeError('FusedAdam: no parameters')
        metas = [getattr(p, '_a4r_flat', None) for p in plist]
        if any(m is None for m in metas):
            raise RuntimeError('FusedAdam: parameters are not bound to a native engine yet (run one forward first); '
                               'there is no eager fallback')
        eng = metas[0][0]
        if any(m[0] is not eng for m in metas):
            raise RuntimeError('FusedAdam: parameters belong to different engines')
        dev = eng.dev
        segs = sorted((m[1], m[1] + m[2], gi) for gi, g in enumerate(self.param_groups) for p in g['params']
                      for m in [p._a4r_flat])
        seg_end = [e for _, e, _ in segs]
        seg_end[-1] = eng.flat_p.numel()                       # alignment padding at the tail
        self._seg_end = torch.tensor(seg_end, dtype=torch.int32, device=dev)
        self._seg_group = torch.tensor([g for _, _, g in segs], dtype=torch.int32, device=dev)
        self._lr_host = None
        self._lr_dev = torch.zeros(len(self.param_groups), dtype=torch.float32, device=dev)
        self._m = torch.zeros_like(eng.flat_p)
        self._v = torch.zeros_like(eng.flat_p)
        covered = sum(m[2] for m in metas)
        if covered != sum(n for _, n in eng.offsets.values()):
            raise RuntimeError('FusedAdam must own every trainable parameter of the engine (the flat buffer is updated as a whole)')
        self._bound = eng
        eng._fused_opt = self
        self._attach_grads()
        self._apply_pending()

    def _attach_grads(self):
        eng = self._bound
        for g in self.param_groups:
            for p in g['params']:
                _, off, n = p._a4r_flat
                view = eng.flat_g[off:off + n].view(p.shape)
                if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                    if p.grad is not None:
                        view.copy_(p.grad)
                    p.grad = view

    def zero_grad(self, set_to_none=False):
        if self._bound is None:
            return super().zero_grad(set_to_none=True)
        self._bound.flat_g.zero_()
        self._bound._flat_clean = True          # the next backward writes straight into flat_g (engine.backward_bound)
        self._attach_grads()

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        if closure is not None:
            raise NotImplementedError('closure')
        if self._bound is None:
            self._bind()
        elif next(p for g in self.param_groups for p in g['params'])._a4r_flat[0] is not self._bound:
            # the model rebuilt its engine (.to(device) / load_state_dict): move the moments over to the new flat buffers
            self._pending = dict(step=self._step, m=self._m, v=self._v)
            self._bind()
        else:
            self._attach_grads()
        eng = self._bound
        lrs = [float(g['lr']) for g in self.param_groups]
        if lrs != self._lr_host:
            self._lr_dev.copy_(torch.tensor(lrs, dtype=torch.float32))
            self._lr_host = lrs
        g0 = self.param_groups[0]
        self._step += 1
        L.adam_step(eng.flat_p, eng.flat_g, self._m, self._v, self._seg_end, self._seg_group, self._lr_dev, self._step,
                    beta1=g0['betas'][0], beta2=g0['betas'][1], eps=g0['eps'], grad_scale=grad_scale)

    def state_dict(self):
        sd = super().state_dict()
        if self._bound is not None:
            sd['a4r_flat'] = dict(step=self._step, m=self._m.clone(), v=self._v.clone())
        return sd

    def load_state_dict(self, sd):
        flat = sd.pop('a4r_flat', None) if isinstance(sd, dict) else None
        super().load_state_dict(sd)
        if flat is not None:
            self._pending = flat

    def _apply_pending(self):
        flat = getattr(self, '_pending', None)
        if flat is not None and self._bound is not None:
            self._step = flat['step']
            self._m.copy_(flat['m'])
            self._v.copy_(flat['v'])
            self._pending = None
