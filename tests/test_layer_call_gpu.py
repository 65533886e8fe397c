"""a4r_encoder_layer_fwd / _bwd (ABI 409, SURVEY 8(b) `encoder_layer_fwd / bwd`): one C call per post-LN encoder layer with serial Houlsby adapters.
The library sequences the same entry points with the same arguments as engine.py's per-launch path: a training step through the layer calls must be
BIT-IDENTICAL to the step with A4R_LAYER_CALL off -- loss, scores, every gradient -- with dropout on and off, with and without the 24-bit residual
stream, on a rectangular batch and on packed titles."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _step(model, items, mask, layer_call, residual, train, host=False):
    import adapter4rec_amd.engine as E
    inner = getattr(model, 'model', model)
    inner.compute_dtype = 'bf16'
    inner.args.residual_dtype = residual
    E.TransRecEngine.LAYER_CALL = layer_call
    inner.invalidate_native()
    for p in model.parameters():
        p.grad = None
    model.to(DEV)
    model.train(train)
    eng = inner._engine()
    eng.step_count = 0                                  # same seed -> same dropout masks in both runs
    calls = {'n': 0}
    from adapter4rec_amd import _lib as L
    real = L.encoder_layer_fwd

    def counted(*a):
        calls['n'] += 1
        return real(*a)
    L.encoder_layer_fwd = counted
    try:
        loss = model(items, mask, DEV) if host else model(items.to(DEV), mask.to(DEV), DEV)
        pos, neg = eng.scores()
        loss.backward()
    finally:
        L.encoder_layer_fwd = real
    out = dict(loss=loss.detach().cpu().clone(), pos=pos.cpu().clone(), neg=neg.cpu().clone(), calls=calls['n'],
               grads={n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.requires_grad})
    model.cpu()
    return out


@pytest.mark.parametrize('residual,train,host', [('bf24', True, False), ('bf16', True, False), ('bf24', False, False), ('bf24', True, True), ('bf20', True, False)])
def test_layer_calls_are_bit_identical_to_the_per_launch_path(residual, train, host):
    import adapter4rec_amd.engine as E
    from base_cases import build_text_case
    model, items, mask = build_text_case('bert', 'GELU', users=2)
    if host:                                            # titles of different lengths: the packed form (offsets) through the same layer call
        S = items.shape[1] // 2
        g = torch.Generator().manual_seed(9)
        lens = torch.randint(5, 25, (items.shape[0],), generator=g)
        col = torch.arange(S)[None, :]
        items = items.clone()
        items[:, :S] = torch.where(col < lens[:, None], items[:, :S], torch.zeros_like(items[:, :S]))
        items[:, S:] = (col < lens[:, None]).long()
    try:
        a = _step(model, items, mask, True, residual, train, host)
        b = _step(model, items, mask, False, residual, train, host)
    finally:
        E.TransRecEngine.LAYER_CALL = True
    assert a['calls'] >= 11 and b['calls'] == 0, (a['calls'], b['calls'])         # the 11 full layers of BERT-base (the 12th runs its second half on the CLS rows)
    # every score bit for bit (the forward has no atomics); the loss is an atomic sum over workgroups of those scores: equal up to its order
    assert torch.equal(a['pos'], b['pos']) and torch.equal(a['neg'], b['neg'])
    assert abs(float(a['loss']) - float(b['loss'])) <= 1e-6 * abs(float(b['loss']))
    for n in a['grads']:
        ga, gb = a['grads'][n], b['grads'][n]
        # weight gradients are fp32 atomic sums over workgroups: equal up to their order (both paths launch the same kernels)
        assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max()) + 1e-12, n


def test_layer_call_refuses_what_it_does_not_cover():
    """A4R_EINVAL (RuntimeError through the binding) for a geometry outside the documented scope; nothing is launched."""
    from adapter4rec_amd import _lib as L
    d = L.EncoderLayer()
    d.M, d.H, d.F, d.S, d.n_heads, d.dh, d.n_items = 256, 768, 3072, 64, 12, 64, 4          # 64 tokens: the long attention kernels are not sequenced here
    x = torch.zeros(256, 768, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError):
        L.encoder_layer_fwd(d, x, x, x)
