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
            old = self._bound
            self._bind()
            self._bound.step_count = max(self._bound.step_count, old.step_count)     # the counter-based dropout stream continues, it does not restart
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

    # -- checkpoint interchange with the reference (Downstream/Text/run.py:481-492, data_utils/utils.py:109-115): torch.optim.Adam's own
    #    layout -- state[param] = {step, exp_avg, exp_avg_sq} -- written and read; the per-parameter tensors are views of the flat moments
    def _state_views(self):
        for g in self.param_groups:
            for p in g['params']:
                _, off, n = p._a4r_flat
                self.state[p] = dict(step=torch.tensor(float(self._step)), exp_avg=self._m[off:off + n].view(p.shape),
                                     exp_avg_sq=self._v[off:off + n].view(p.shape))

    def state_dict(self):
        if self._bound is not None:
            self._state_views()
        sd = super().state_dict()
        if self._bound is not None:
            sd['a4r'] = dict(engine_step_count=int(self._bound.step_count))      # the counter-based dropout stream resumes where it stopped
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)                                          # the caller's dict is not modified
        extra, legacy = sd.pop('a4r', None), sd.pop('a4r_flat', None)
        super().load_state_dict(sd)                            # a reference checkpoint's exp_avg / exp_avg_sq / step land in self.state
        self._pending = dict(extra=extra, legacy=legacy, late=self._bound is not None)
        if extra:                                               # the engine may not exist yet (built at the first forward): leave the
            for g in self.param_groups:                         # dropout counter on the parameters, TransRecEngine picks it up
                for p in g['params']:
                    if getattr(p, '_a4r_flat', None) is not None:
                        p._a4r_flat[0].step_count = int(extra.get('engine_step_count', 0))
                    else:
                        p._a4r_resume_step = int(extra.get('engine_step_count', 0))
        if self._bound is not None:
            self._apply_pending()

    def _apply_pending(self):
        pend = getattr(self, '_pending', None)
        if pend is None or self._bound is None:
            return
        self._pending = None
        if 'm' in pend:                                        # moments carried over an engine rebuild (step())
            self._step = pend['step']
            self._m.copy_(pend['m'])
            self._v.copy_(pend['v'])
            return
        if pend.get('legacy') is not None:                     # round-1 checkpoints of this package
            self._step = pend['legacy']['step']
            self._m.copy_(pend['legacy']['m'])
            self._v.copy_(pend['legacy']['v'])
        else:
            for g in self.param_groups:
                for p in g['params']:
                    st = self.state.get(p)
                    if st and 'exp_avg' in st:
                        _, off, n = p._a4r_flat
                        self._m[off:off + n].copy_(st['exp_avg'].reshape(-1))
                        self._v[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
                        self._step = int(float(st['step']))
            self._state_views()
        if pend.get('extra') and pend.get('late'):             # engine existed when the state was loaded; a lazily built
            self._bound.step_count = int(pend['extra'].get('engine_step_count', self._bound.step_count))   # one read p._a4r_resume_step
