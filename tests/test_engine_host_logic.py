"""CPU: the engine's HOST logic (buffer plumbing, launch order, gradient routing, FusedAdam binding) driven through
tests/sim_lib.py -- a torch restatement of the C-ABI kernels' documented semantics -- and checked against the
reference's golden vectors.  This does not test the HIP kernels (tests/test_kernels_gpu.py, test_engine_gpu.py do);
it keeps the Python orchestration honest in the GPU-less build container."""
import copy
import numpy as np
import pytest
import torch

import sim_lib
from golden_util import strip
import test_engine_gpu as TG


@pytest.fixture
def simulated(monkeypatch):
    import adapter4rec_amd.engine as E
    import adapter4rec_amd.optim as O
    monkeypatch.setattr(E, 'L', sim_lib)
    monkeypatch.setattr(O, 'L', sim_lib)
    monkeypatch.setattr(E.TransRecEngine, '_require_device', lambda self, p0: None)


def build_cpu(name):
    import adapter4rec_amd.inject as I
    from adapter4rec_amd.model import BertBackbone, Model, ModelCPC
    from golden_util import load_variant
    sd, cfg, fx, trainable, (items, mask), base = load_variant(name)
    args = TG.make_args(compute_dtype='fp32', **TG.ARGS[name])
    geom = dict(TG.GEOM)
    if name.startswith('roberta'):
        geom.update(max_position_embeddings=42, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1, model_type='roberta')
    model = (ModelCPC if args.arch == 'cpc' else Model)(args, 200, True, BertBackbone(geom))
    I.freeze_all(model)
    root = I.inject_adapters(model, args)
    root.load_state_dict({str(k): sd[strip(str(k))] for k in fx['all_keys']}, strict=True)
    root.eval()
    return root, args, fx, items, mask


@pytest.mark.parametrize('name', list(TG.ARGS))
def test_host_logic_forward_backward(simulated, name):
    root, args, fx, items, mask = build_cpu(name)
    loss = root(items, mask, 'cpu')
    loss.backward()
    assert abs(loss.item() - float(fx['loss'])) < 1e-4
    inner = getattr(root, 'model', root)
    np.testing.assert_allclose(inner.bert_encoder(items).numpy(), fx['input_embs_all'], atol=1e-4, rtol=0)
    e = torch.from_numpy(fx['input_embs_all']).view(-1, 21, 2, 64)
    prec = inner.user_encoder(e[:, :-1, 0].contiguous(), mask, 'cpu')
    np.testing.assert_allclose(prec.numpy(), fx['prec_vec'], atol=1e-4, rtol=0)
    params = dict(root.named_parameters())
    for k in fx['trainable']:
        k = str(k)
        ref = fx['grad/' + k]
        np.testing.assert_allclose(params[k].grad.numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=k)


@pytest.mark.parametrize('name', ['houlsby', 'houlsby_cpc', 'roberta_cpc_pfeiffer'])
def test_host_logic_unused_item_slots_not_encoded(simulated, monkeypatch, name):
    """The compact item batch forced on (A4R_SKIP_UNUSED_ITEMS=2; CPC compacts by itself): the item slots Model.forward / ModelCPC.forward never
    read are not encoded, and loss and every gradient still equal the reference's fixture (same checks as test_host_logic)."""
    monkeypatch.setenv('A4R_SKIP_UNUSED_ITEMS', '2')
    root, args, fx, items, mask = build_cpu(name)
    inner = getattr(root, 'model', root)
    assert inner._engine()._kept_rows(mask.shape[0]) is not None
    loss = root(items, mask, 'cpu')
    loss.backward()
    assert abs(loss.item() - float(fx['loss'])) < 1e-4
    params = dict(root.named_parameters())
    for k in fx['trainable']:
        k = str(k)
        ref = fx['grad/' + k]
        np.testing.assert_allclose(params[k].grad.numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=k)


def short_title_case(name, device='cpu', n_tok=16):
    """A fixture's model on a batch whose titles are all cut to n_tok tokens (ids and attention mask zeroed behind them): handed over on the host, the step
    runs on n_tok tokens per item instead of 30 (engine.py: _set_S).  The checker is the oracle on the same batch at the FULL title length."""
    from oracle import ref_cpu as R
    from golden_util import load_variant
    root, args, fx, items, mask = build_cpu(name)
    sd, cfg, *_ = load_variant(name)
    items = items.clone()
    S = items.shape[1] // 2
    items[:, n_tok:S] = 1 if name.startswith('roberta') else 0
    items[:, S + n_tok:] = 0
    names = [str(k) for k in fx['trainable']]
    out, grads = R.loss_and_grads(sd, names, items, mask, cfg)
    return root.to(device), items, mask, names, out, grads


@pytest.mark.parametrize('name', ['houlsby', 'roberta_cpc_pfeiffer', 'pfeiffer'])
def test_host_logic_short_titles_run_on_fewer_tokens(simulated, name):
    root, items, mask, names, out, grads = short_title_case(name)
    inner = getattr(root, 'model', root)
    loss = root(items, mask, 'cpu')
    assert inner._engine()._ctx['S'] == 16
    loss.backward()
    assert abs(loss.item() - float(out['loss'].detach())) < 1e-4
    params = dict(root.named_parameters())
    for n in names:
        ref = grads[n].numpy()
        np.testing.assert_allclose(params[n].grad.numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=n)
    emb = inner.bert_encoder(items)                 # inference afterwards: back on the full title length
    assert inner._engine().S == items.shape[1] // 2 and emb.shape[0] == items.shape[0]


@pytest.mark.parametrize('name', ['houlsby', 'roberta_cpc_pfeiffer'])
def test_host_logic_non_prefix_mask_keeps_every_attended_token(simulated, name):
    """the per-step title length is bounded by the LAST attended column, not by the count of mask ones.  Masks with a hole and with one
    token attended far behind the others (11 ones, the last in column 20): the step must keep 22 columns, and loss / gradients must equal the
    oracle on the full 30-token rows."""
    from oracle import ref_cpu as R
    from golden_util import load_variant
    root, args, fx, items, mask = build_cpu(name)
    sd, cfg, *_ = load_variant(name)
    items = items.clone()
    S = items.shape[1] // 2
    items[:, S:] = 0
    items[:, S:S + 10] = 1
    items[:, S + 3] = 0                  # a hole
    items[1::3, S + 20] = 1              # one straggler per third item, far behind the prefix
    names = [str(k) for k in fx['trainable']]
    out, grads = R.loss_and_grads(sd, names, items, mask, cfg)
    inner = getattr(root, 'model', root)
    loss = root(items, mask, 'cpu')
    assert inner._engine()._ctx['S'] == 22
    loss.backward()
    assert abs(loss.item() - float(out['loss'].detach())) < 1e-4
    params = dict(root.named_parameters())
    for n in names:
        ref = grads[n].numpy()
        np.testing.assert_allclose(params[n].grad.numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=n)


def ragged_title_case(name, device='cpu', seed=5):
    """A fixture's model on a batch whose titles have DIFFERENT lengths (3 .. 20 attended tokens, ids and mask zeroed behind them; RoBERTa: pad id 1):
    handed over on the host the titles are PACKED -- item i runs on its own token rows (engine.py: train_forward, self._pk; a4r_attn_t.offsets).
    The checker is the oracle on the same batch in the rectangular 30-token layout."""
    from oracle import ref_cpu as R
    from golden_util import load_variant
    root, args, fx, items, mask = build_cpu(name)
    sd, cfg, *_ = load_variant(name)
    items = items.clone()
    S = items.shape[1] // 2
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(3, 21, (items.shape[0],), generator=g)
    col = torch.arange(S)[None, :]
    items[:, :S] = torch.where(col < lens[:, None], items[:, :S], torch.full_like(items[:, :S], 1 if name.startswith('roberta') else 0))
    items[:, S:] = (col < lens[:, None]).long()
    names = [str(k) for k in fx['trainable']]
    out, grads = R.loss_and_grads(sd, names, items, mask, cfg)
    return root.to(device), items, mask, names, out, grads, int(lens.sum())


@pytest.mark.parametrize('name', ['houlsby', 'roberta_cpc_pfeiffer', 'pfeiffer', 'houlsby_parallel'])
def test_host_logic_titles_of_different_lengths_are_packed(simulated, name, monkeypatch):
    monkeypatch.setenv('A4R_SKIP_UNUSED_ITEMS', '1')
    root, items, mask, names, out, grads, n_tok = ragged_title_case(name)
    inner = getattr(root, 'model', root)
    loss = root(items, mask, 'cpu')
    eng = inner._engine()
    pk = eng._ctx['pk']
    assert pk is not None and pk['Mtok'] <= n_tok and eng._ctx['M'] == -(-pk['Mtok'] // 256) * 256 < items.shape[0] * eng._ctx['S']
    loss.backward()
    assert abs(loss.item() - float(out['loss'].detach())) < 1e-4
    params = dict(root.named_parameters())
    for n in names:
        ref = grads[n].numpy()
        np.testing.assert_allclose(params[n].grad.numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=n)
    emb = inner.bert_encoder(items)                 # inference afterwards: the rectangular layout again
    assert eng._pk is None and emb.shape[0] == items.shape[0]


@pytest.mark.parametrize('name', ['houlsby', 'compacter'])
def test_host_logic_fused_adam(simulated, name):
    from adapter4rec_amd.inject import optimizer_groups
    from adapter4rec_amd.optim import FusedAdam
    root, args, fx, items, mask = build_cpu(name)
    opt = FusedAdam(optimizer_groups(root, args))
    params = dict(root.named_parameters())
    losses = []
    for s in range(3):
        opt.zero_grad()
        loss = root(items, mask, 'cpu')
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if s in (0, 2):
            for k in fx['trainable']:
                k = str(k)
                np.testing.assert_allclose(params[k].detach().numpy(), fx[f'adam{s + 1}/' + k], rtol=2e-4, atol=2e-7, err_msg=k)
    np.testing.assert_allclose(losses, fx['adam_losses'], atol=1e-4, rtol=0)


@pytest.mark.parametrize('name', ['houlsby', 'roberta_cpc_pfeiffer', 'compacter'])
def test_host_logic_bf16_mode(simulated, name):
    """bf16 storage for the item encoder, fp32 for the SASRec side: mixed-dtype plumbing (pack tables per dtype, mixed
    GEMM in/out types) must keep the step within bf16 rounding of the fp32 oracle on well-conditioned weights."""
    from golden_util import load_variant
    from oracle import ref_cpu as R
    root, args, fx, items, mask = build_cpu(name)
    sd, cfg, *_ = load_variant(name)
    sd = TG.condition(sd)
    inner = getattr(root, 'model', root)
    inner.compute_dtype = 'bf16'
    root.load_state_dict({str(k): sd[strip(str(k))] for k in fx['all_keys']}, strict=True)
    tr = [strip(str(k)) for k in fx['trainable']]
    out, grads = R.loss_and_grads(sd, tr, items, mask, cfg)
    loss = root(items, mask, 'cpu')
    loss.backward()
    assert abs(loss.item() - float(out['loss'].detach())) < 2e-2
    params = dict(root.named_parameters())
    for k in fx['trainable']:
        k = str(k)
        ref = grads[strip(k)].numpy()
        assert np.abs(params[k].grad.numpy() - ref).max() <= 0.15 * np.abs(ref).max() + 1e-9, k      # (bf16 noise: 0.11 - 0.14 by which roundings the forward has)


@pytest.mark.parametrize('name', ['houlsby', 'pfeiffer', 'roberta_cpc_pfeiffer', 'prompt'])
def test_host_logic_residual_fp32(simulated, name):
    """--residual_dtype fp32 through every sub-layer form of the text tower (one-launch serial adapter, un-adapted sub-layer and Pfeiffer's FFN
    half through a4r_ln_fwd_sum): the step stays inside the bf16 bounds against the fp32 oracle, the fp32 twins are consumed (embeddings
    differ from the bf16-stream run) and are at least as close to the oracle."""
    from golden_util import load_variant
    from oracle import ref_cpu as R
    root, args, fx, items, mask = build_cpu(name)
    sd, cfg, *_ = load_variant(name)
    sd = TG.condition(sd)
    inner = getattr(root, 'model', root)
    inner.compute_dtype = 'bf16'
    root.load_state_dict({str(k): sd[strip(str(k))] for k in fx['all_keys']}, strict=True)
    tr = [strip(str(k)) for k in fx['trainable']]
    out, grads = R.loss_and_grads(sd, tr, items, mask, cfg)
    res = {}
    for rd in ('bf16', 'fp32', 'bf24', 'bf20'):      # bf24 (round 6; bf20: a nibble per element): the same twins as one byte per element, on the one-launch serial adapter sub-layers
        inner.args.residual_dtype = rd
        inner.invalidate_native()
        for p in root.parameters():
            p.grad = None
        loss = root(items, mask, 'cpu')
        loss.backward()
        with torch.no_grad():
            emb = inner.bert_encoder(items)
        params = dict(root.named_parameters())
        rel = {str(k): float(np.abs(params[str(k)].grad.numpy() - grads[strip(str(k))].numpy()).max() / (np.abs(grads[strip(str(k))].numpy()).max() + 1e-12))
               for k in fx['trainable']}
        # RELU-gated down-projection gradients under CPC (one scored position per user) rest on a handful of token rows: a relu' that flips
        # under bf16 rounding moves them by 0.07 - 0.25 of their max, either way, with any change of the roundings upstream -- bounded apart
        gated = [v for k, v in rel.items() if 'fc_down' in k]
        worst = max(v for k, v in rel.items() if 'fc_down' not in k or cfg['arch'] != 'cpc')
        assert not gated or max(gated) < 0.3, max(gated)
        res[rd] = (abs(loss.item() - float(out['loss'].detach())), emb, float((emb - out['input_embs_all'].detach()).double().pow(2).mean().sqrt()), worst)
    print(name, {k: (round(v[0], 5), round(v[2], 6), round(v[3], 4)) for k, v in res.items()})
    assert res['fp32'][0] < 2e-2 and res['fp32'][3] < 0.15, (res['fp32'][::3], res['bf16'][::3])
    assert not torch.equal(res['fp32'][1], res['bf16'][1])
    assert res['fp32'][2] <= 1.05 * res['bf16'][2], (res['fp32'][2], res['bf16'][2])
    assert res['bf24'][0] < 2e-2 and res['bf24'][3] < 0.15, res['bf24'][::3]
    assert res['bf24'][2] <= 1.05 * res['bf16'][2], (res['bf24'][2], res['bf16'][2])
    assert res['bf20'][0] < 2e-2 and res['bf20'][3] < 0.15 and res['bf20'][2] <= 1.05 * res['bf16'][2], res['bf20'][::3]
    if name == 'houlsby':                            # every sub-layer on the one-launch kernel: the byte planes are consumed, and buy what the fp32 twins buy
        assert not torch.equal(res['bf24'][1], res['bf16'][1])
        assert res['bf24'][2] <= 1.02 * res['fp32'][2] + 1e-6, (res['bf24'][2], res['fp32'][2])
        assert not torch.equal(res['bf20'][1], res['bf16'][1])          # (bf20 and bf24 may coincide after the two layers' bf16 roundings at this geometry)
        assert res['bf20'][2] <= 1.05 * res['fp32'][2] + 1e-6, (res['bf20'][2], res['fp32'][2])


def build_lora_cpu(dtype='fp32'):
    """LoRA has no reference fixture (loralib is absent): weights are seeded here and the oracle is the only checker."""
    import adapter4rec_amd.inject as I
    from adapter4rec_amd.model import BertBackbone, Model
    from golden_util import load_variant
    sd, cfg, fx, trainable, (items, mask), base = load_variant('finetune_all')       # un-adapted base weights
    args = TG.make_args(compute_dtype=dtype, adapter_type='lora', bert_adapter_down_size=8, adapter_down_size=4)
    model = Model(args, 200, True, BertBackbone(dict(TG.GEOM)))
    model.load_state_dict({str(k): sd[strip(str(k))] for k in fx['all_keys']}, strict=True)
    I.freeze_all(model)
    torch.manual_seed(7)
    model = I.inject_adapters(model, args)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if 'lora_B' in n:
                p.normal_(std=0.05)                    # zeros would hide dA and half of the forward
    model.eval()
    osd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ocfg = dict(cfg, adapter_type='lora', lora_r_bert=8, lora_r_sasrec=4)
    return model, args, osd, ocfg, items, mask


def lora_pin_step(dev, tower='text'):
    """The LoRA model on lora_pin_case's weights (W = W_base - B A / r): loss / embeddings vs the IMPORTED reference's numbers, dA / dB /
    dbias vs the chain rule through the reference's own dL/dW (golden_util.lora_pin_case) -- a7 pinned without loralib."""
    import adapter4rec_amd.inject as I
    from golden_util import lora_pin_case
    sd, cfg, (items, mask), pin, expect = lora_pin_case(tower)
    if tower == 'text':
        from adapter4rec_amd.model import BertBackbone, Model
        args = TG.make_args(compute_dtype='fp32', adapter_type='lora', bert_adapter_down_size=8, adapter_down_size=4)
        model = Model(args, 200, True, BertBackbone(dict(TG.GEOM)))
        inj = I.inject_adapters
    else:
        import test_engine_cv as TC
        from adapter4rec_amd.cv import Model, ViTForImageClassification
        from adapter4rec_amd.cv.inject import inject_adapters as inj
        args = TC.make_args(adapter_type='lora', lora_r=8, lora_r_sasrec=4, compute_dtype='fp32')
        model = Model(args, 60, True, ViTForImageClassification(TC.GEOM))
    I.freeze_all(model)
    model = inj(model, args)
    # (the reference's SASRec w_Q / w_V are bias-free Linears; the lora.Linear that replaces them has a bias: zero = the base layer)
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.endswith(('w_Q.bias', 'w_V.bias')) for k in missing.missing_keys), missing
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n in missing.missing_keys:
                p.zero_()
    model.eval().to(dev)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert set(expect) <= set(names), sorted(set(expect) - set(names))
    loss = model(items.to(dev), mask.to(dev), dev)
    loss.backward()
    assert abs(loss.item() - float(pin['loss'])) < 1e-4
    params = dict(model.named_parameters())
    for n, ref in expect.items():
        ref = ref.numpy()
        np.testing.assert_allclose(params[n].grad.cpu().numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=n)


def test_host_logic_lora_pinned(simulated):
    lora_pin_step('cpu', 'text')


def test_host_logic_lora(simulated):
    from oracle import ref_cpu as R
    model, args, osd, ocfg, items, mask = build_lora_cpu()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert any('lora_A' in n for n in names) and any(n.endswith('query.bias') for n in names)
    out, grads = R.loss_and_grads(osd, names, items, mask, ocfg)
    loss = model(items, mask, 'cpu')
    loss.backward()
    assert abs(loss.item() - float(out['loss'].detach())) < 1e-4
    params = dict(model.named_parameters())
    for n in names:
        ref = grads[n].numpy()
        np.testing.assert_allclose(params[n].grad.numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=n)


def build_finetune_all(device='cpu', dtype='fp32', ln_only=False):
    """--fine_tune_to all (run.py:366-371: nothing frozen, no adapters) -- the Pretraining/ configuration -- or, with ln_only,
    adapters off + --finetune_layernorm (every LayerNorm incl. the embedding one trainable)."""
    from adapter4rec_amd.model import BertBackbone, Model
    from golden_util import load_variant
    sd, cfg, fx, trainable, (items, mask), base = load_variant('finetune_all')
    args = TG.make_args(compute_dtype=dtype, adapter_type='none', adding_adapter_to='None')
    model = Model(args, 200, True, BertBackbone(dict(TG.GEOM)))
    model.load_state_dict({str(k): sd[strip(str(k))] for k in fx['all_keys']}, strict=True)
    for n, p in model.named_parameters():
        if ln_only:
            p.requires_grad = ('LayerNorm' in n or 'layer_norm' in n)
        else:
            p.requires_grad = 'pooler' not in n            # as the fixture generator (pooler never receives a gradient)
    model.eval()
    return model.to(device), args, sd, cfg, fx, items.to(device), mask.to(device)


@pytest.mark.parametrize('ln_only', [False, True])
def test_host_logic_finetune_all(simulated, ln_only):
    from oracle import ref_cpu as R
    model, args, sd, cfg, fx, items, mask = build_finetune_all(ln_only=ln_only)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    out, grads = R.loss_and_grads(sd, names, items, mask, cfg)
    loss = model(items, mask, 'cpu')
    loss.backward()
    assert abs(loss.item() - float(fx['loss'])) < 1e-4
    params = dict(model.named_parameters())
    for n in names:                                           # every tensor against the oracle's autograd ...
        ref = grads[n].numpy()
        np.testing.assert_allclose(params[n].grad.numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=n)
    if not ln_only:
        for k in fx.files:                                    # ... and the ones the reference itself recorded
            if k.startswith('grad/'):
                ref = fx[k]
                np.testing.assert_allclose(params[k[5:]].grad.numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=k)


def test_host_logic_bound_backward_path(simulated):
    """The public path once FusedAdam is bound (engine.backward_bound): p.grad are views of the flat gradient buffer and are
    filled in place -- same numbers as the per-parameter autograd path, a scaled loss scales them (grad_output stays a device
    scalar read by the head kernel), and a second backward without zero_grad accumulates."""
    from adapter4rec_amd.inject import optimizer_groups
    from adapter4rec_amd.optim import FusedAdam
    root, args, fx, items, mask = build_cpu('houlsby')
    params = dict(root.named_parameters())
    root(items, mask, 'cpu').backward()                              # unbound: per-parameter gradients through autograd
    ref = {k: params[k].grad.detach().clone() for k in map(str, fx['trainable'])}
    opt = FusedAdam(optimizer_groups(root, args), lr=0.0)
    for g in opt.param_groups:
        g['lr'] = 0.0
    opt.step()                                                       # binds (lr 0: parameters unchanged)
    eng = getattr(root, 'model', root)._engine()
    assert eng._fused_opt is opt
    opt.zero_grad()
    (0.5 * root(items, mask, 'cpu')).backward()
    for k, r in ref.items():
        assert params[k].grad.data_ptr() == eng.flat_g[eng.offsets[id(params[k])][0]:].data_ptr()
        np.testing.assert_allclose(params[k].grad.numpy(), 0.5 * r.numpy(), atol=1e-7 + 1e-5 * r.abs().max().item(), rtol=0, err_msg=k)
    root(items, mask, 'cpu').backward()                              # accumulation: 0.5 g + g
    for k, r in ref.items():
        np.testing.assert_allclose(params[k].grad.numpy(), 1.5 * r.numpy(), atol=1e-7 + 1e-5 * r.abs().max().item(), rtol=0, err_msg=k)


@pytest.mark.parametrize('variant', ['houlsby', 'compacter'])
def test_host_logic_eval_item_sweep_in_fp32_snapshot(simulated, variant):
    """--eval_compute_dtype fp32 under bf16 training: the item sweep runs on a forward-only fp32 snapshot engine of the CURRENT
    weights; it equals the fp32 engine's embeddings and leaves the training engine (which owns the parameters' flat buffer) alone.
    compacter: the snapshot's PHM tensors are frozen -- their effective matrices still come from a4r_phm_build (round 6: no eager bmm)."""
    root, args, fx, items, mask = build_cpu(variant)
    inner = getattr(root, 'model', root)
    ref = inner.bert_encoder(items).clone()                          # fp32 engine
    inner.compute_dtype = 'bf16'
    inner.invalidate_native()
    eng = inner._engine()
    flat = eng.flat_p
    with torch.no_grad():
        for p in eng.trainable_params:                               # "training" moved the weights: the snapshot must see the new values
            p.mul_(1.5)
    inner.compute_dtype = 'fp32'
    inner.invalidate_native()
    ref2 = inner.bert_encoder(items).clone()
    assert (ref2 - ref).abs().max() > 1e-4
    inner.compute_dtype = 'bf16'
    inner.invalidate_native()
    eng = inner._engine()
    got = inner.item_encoder_in('fp32')(items)
    np.testing.assert_allclose(got.numpy(), ref2.numpy(), atol=1e-6, rtol=0)
    assert inner._engine() is eng and all(p.requires_grad for p in eng.trainable_params)
    bf = inner.bert_encoder(items)
    assert 0 < (bf - ref2).abs().max() < 5e-2                        # the bf16 engine itself still answers, in bf16


def test_host_logic_optimizer_checkpoint_interchange(simulated):
    """Resume fidelity: FusedAdam writes and reads torch.optim.Adam's own state layout (state[param] = step / exp_avg /
    exp_avg_sq, what a reference checkpoint holds, run.py:481-492), continues bit-for-bit after a save / load, restores the
    engine's dropout counter, and does not modify the dict it is given."""
    from adapter4rec_amd.inject import optimizer_groups
    from adapter4rec_amd.optim import FusedAdam

    def run(n, opt, root, items, mask):
        for _ in range(n):
            opt.zero_grad()
            root(items, mask, 'cpu').backward()
            opt.step()

    root, args, fx, items, mask = build_cpu('houlsby')
    opt = FusedAdam(optimizer_groups(root, args))
    run(2, opt, root, items, mask)
    sd_opt = copy.deepcopy(opt.state_dict())            # what torch.save would hold (state_dict itself returns live views, like Adam's)
    sd_model = {k: v.clone() for k, v in root.state_dict().items()}
    # the layout torch.optim.Adam itself would have written
    names = [n for n, p in root.named_parameters() if p.requires_grad]
    st = sd_opt['state']
    assert len(st) == len(names) and all(set(v) >= {'step', 'exp_avg', 'exp_avg_sq'} for v in st.values())
    assert float(next(iter(st.values()))['step']) == 2.0 and sd_opt['a4r']['engine_step_count'] == 2
    adam = torch.optim.Adam(optimizer_groups(root, args))
    adam.load_state_dict({k: v for k, v in sd_opt.items() if k != 'a4r'})                 # a plain Adam accepts it
    run(1, opt, root, items, mask)
    after3 = {k: v.clone() for k, v in root.state_dict().items()}
    # resume in a fresh model + optimizer from the saved pair
    root2, args2, _, _, _ = build_cpu('houlsby')
    root2.load_state_dict(sd_model)
    opt2 = FusedAdam(optimizer_groups(root2, args2))
    keys_before = set(sd_opt)
    opt2.load_state_dict(sd_opt)
    assert set(sd_opt) == keys_before                                                      # caller's dict untouched
    run(1, opt2, root2, items, mask)
    assert getattr(root2, 'model', root2)._engine().step_count == 3
    for k, v in after3.items():
        torch.testing.assert_close(root2.state_dict()[k], v, rtol=0, atol=0)


@pytest.mark.parametrize('adapter_type', ['houslby', 'pfeiffer'])
def test_host_logic_text_fp8_encoder(simulated, adapter_type):
    """The fp8 wiring of the post-LN text tower (e4m3 rows handed from layer to layer, FFN dgrad chain) through the CPU restatement."""
    TG._text_fp8_case('cpu', adapter_type)


@pytest.mark.parametrize('name', ['kadapter', 'prompt', 'houlsby_parallel', 'compacter', 'pfeiffer_ver2', 'houlsby_cpc', 'roberta_prompt'])
def test_host_logic_text_fp8_every_placement(simulated, name):
    """--compute_dtype fp8 on every adapter placement of the text tower (tiny geometry: only the FFN-up operand has a 256-tile shape, the other
    GEMMs of the same blocks stay bf16): the wiring runs, the loss stays next to the bf16 run's and every gradient is finite and close."""
    root, args, fx, items, mask = build_cpu(name)
    inner = getattr(root, 'model', root)
    res = {}
    for dt in ('bf16', 'fp8'):
        inner.compute_dtype = dt
        inner.invalidate_native()
        for p in root.parameters():
            p.grad = None
        loss = root(items, mask, 'cpu')
        loss.backward()
        res[dt] = (loss.item(), {n: p.grad.clone() for n, p in root.named_parameters() if p.requires_grad})
    eng = inner._engine()
    assert eng.fp8 and all(getattr(b, 'wi8', None) is not None for b in eng.bert_blocks)
    assert abs(res['fp8'][0] - res['bf16'][0]) < 0.05 * max(1.0, abs(res['bf16'][0]))
    for n, g in res['bf16'][1].items():
        g8 = res['fp8'][1][n]
        assert torch.isfinite(g8).all(), n
        assert float((g8 - g).abs().max()) <= 0.5 * float(g.abs().max()) + 1e-6, n


@pytest.mark.parametrize('attrs', [('title', 'abstract'), ('title', 'abstract', 'body'), ('body',)])
def test_host_logic_news_attributes(simulated, attrs):
    """--news_attributes with more than the title (encoders.py:62-99: every attribute through the one Text_Encoder, item vector = their mean): the
    engine stacks the attributes as extra items at the longest attribute's length.  Toy geometry, kernels simulated: loss and every gradient against
    the CPU oracle, through the training entry point with the rows on the host and through the inference entry point."""
    import adapter4rec_amd.inject as I
    from adapter4rec_amd.model import BertBackbone, Model
    from oracle import ref_cpu as R
    torch.manual_seed(7)
    args = TG.make_args(compute_dtype='fp32', **TG.ARGS['houlsby'])
    args.news_attributes, args.num_words_title, args.num_words_abstract, args.num_words_body = list(attrs), 12, 20, 16
    model = Model(args, 200, True, BertBackbone(dict(TG.GEOM)))
    I.freeze_all(model)
    root = I.inject_adapters(model, args)
    with torch.no_grad():
        for n, p in root.named_parameters():
            if p.requires_grad:
                p.add_(0.02 * torch.randn_like(p))
    root.eval()
    lens = [w for a, w in zip(('title', 'abstract', 'body'), (12, 20, 16)) if a in attrs]
    g = torch.Generator().manual_seed(3)
    B, Lq, width = 2, args.max_seq_len + 1, 2 * sum(lens)
    ids = torch.zeros(B, Lq, 2, width, dtype=torch.int64)
    mask = torch.zeros(B, Lq - 1)
    for u, n in enumerate((Lq, 6)):
        for slot in range(Lq - n, Lq):
            for side in range(2):
                if side == 1 and slot == Lq - 1:
                    continue
                st = 0
                for w in lens:
                    ln = int(torch.randint(3, w + 1, (1,), generator=g))
                    ids[u, slot, side, st:st + ln] = torch.randint(5, 90, (ln,), generator=g)
                    ids[u, slot, side, st + w:st + w + ln] = 1
                    st += 2 * w
        mask[u, Lq - n:] = 1
    items = ids.view(-1, width)
    sd = {strip(k): v.detach().clone() for k, v in root.state_dict().items()}
    names = [n for n, p in root.named_parameters() if p.requires_grad]
    cfg = dict(R.DEFAULT_CFG, **TG.CFG['houlsby']) if hasattr(TG, 'CFG') else dict(R.DEFAULT_CFG)
    cfg.update(bert_heads=TG.GEOM['num_attention_heads'], news_attributes=list(attrs), num_words_title=12, num_words_abstract=20, num_words_body=16,
               max_seq_len=args.max_seq_len, embedding_dim=args.embedding_dim, sasrec_heads=args.num_attention_heads,
               adapter_activation=args.adapter_activation)
    out, grads = R.loss_and_grads(sd, [strip(n) for n in names], items, mask, cfg)
    loss = root(items, mask, 'cpu')
    loss.backward()
    assert abs(loss.item() - float(out['loss'].detach())) < 1e-4
    params = dict(root.named_parameters())
    for n in names:
        ref = grads[strip(n)].numpy()
        np.testing.assert_allclose(params[n].grad.numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=n)
    real = (items != 0).any(1)
    emb = getattr(root, 'model', root).bert_encoder(items)
    np.testing.assert_allclose(emb[real].numpy(), out['input_embs_all'].detach()[real].numpy(), atol=1e-4, rtol=0)


@pytest.mark.parametrize('max_len', [33, 45])
def test_host_logic_long_histories(simulated, max_len):
    """--max_seq_len above 32 (parameters.py:29): the user tower's blocks run the causal, key-masked form of the long attention kernels
    (model/user_encoders.py:20-27: att_mask = log_mask & tril).  Toy geometry, kernels simulated: loss and every gradient against the CPU oracle."""
    import adapter4rec_amd.inject as I
    from adapter4rec_amd.model import BertBackbone, Model
    from oracle import ref_cpu as R
    torch.manual_seed(8)
    args = TG.make_args(compute_dtype='fp32', **TG.ARGS['houlsby'])
    args.max_seq_len = max_len
    model = Model(args, 200, True, BertBackbone(dict(TG.GEOM)))
    I.freeze_all(model)
    root = I.inject_adapters(model, args)
    with torch.no_grad():
        for n, p in root.named_parameters():
            if p.requires_grad:
                p.add_(0.02 * torch.randn_like(p))
    root.eval()
    W = args.num_words_title
    g = torch.Generator().manual_seed(4)
    B, Lq = 3, max_len + 1
    ids = torch.zeros(B, Lq, 2, 2 * W, dtype=torch.int64)
    mask = torch.zeros(B, Lq - 1)
    for u, n in enumerate((Lq, 35, 3)):
        for slot in range(Lq - n, Lq):
            for side in range(2):
                if side == 1 and slot == Lq - 1:
                    continue
                ln = int(torch.randint(3, W + 1, (1,), generator=g))
                ids[u, slot, side, :ln] = torch.randint(5, 90, (ln,), generator=g)
                ids[u, slot, side, W:W + ln] = 1
        mask[u, Lq - n:] = 1
    items = ids.view(-1, 2 * W)
    sd = {strip(k): v.detach().clone() for k, v in root.state_dict().items()}
    names = [n for n, p in root.named_parameters() if p.requires_grad]
    cfg = dict(R.DEFAULT_CFG)
    cfg.update(bert_heads=TG.GEOM['num_attention_heads'], num_words_title=W, max_seq_len=max_len, embedding_dim=args.embedding_dim,
               sasrec_heads=args.num_attention_heads, adapter_activation=args.adapter_activation)
    out, grads = R.loss_and_grads(sd, [strip(n) for n in names], items, mask, cfg)
    loss = root(items, mask, 'cpu')
    loss.backward()
    assert abs(loss.item() - float(out['loss'].detach())) < 1e-4
    params = dict(root.named_parameters())
    for n in names:
        ref = grads[strip(n)].numpy()
        np.testing.assert_allclose(params[n].grad.numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=n)


def test_host_logic_news_attributes_vs_reference_golden(simulated):
    """the same fixture through the simulated kernels (CPU): the engine's attribute stacking, the mean and its backward"""
    root, args, fx, items, mask = TG.build_multi_attr('cpu')
    loss = root(items, mask, 'cpu')
    loss.backward()
    assert abs(loss.item() - float(fx['loss'])) < 1e-4
    params = dict(root.named_parameters())
    for k in fx['trainable']:
        ref = fx['grad/' + str(k)]
        np.testing.assert_allclose(params[str(k)].grad.numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=str(k))

