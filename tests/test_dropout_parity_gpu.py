"""The TRAINING-mode step (dropout on: the mode bench.py times) against the CPU oracle under IDENTICAL masks.

The HIP kernels regenerate every keep mask from (seed, site, element index) (csrc/a4r_common.h: dropout_keep); oracle/dropout_masks.py restates
that stream and each kernel family's index, oracle/ref_cpu.py multiplies by those masks at the reference's dropout sites (HF BertEmbeddings /
BertSelfAttention / BertSelfOutput / BertOutput; Downstream/Text/model/modules.py:27,40,70,104; model.py:292-297).  Tolerances: fp32 instantiation
loss 1e-4 abs, every trainable gradient 1e-4 of its tensor's max (north_star); bf16 the bounds of tests/test_engine_gpu.py::test_step_bf16_bound.
First the masks themselves are pinned against the library through its public ops (a4r_dropout_apply; a4r_attn_fwd on Q = K = 0, V = one-hot).
"""
import numpy as np
import pytest
import torch

from golden_util import strip

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('M,N,p,site', [(256, 128, 0.1, 17), (384, 768, 0.1, 999), (128, 64, 0.5, 4000), (256, 3072, 0.25, 34)])
def test_row_masks_equal_the_librarys(M, N, p, site):
    from adapter4rec_amd import _lib as L
    from oracle.dropout_masks import DropoutStream
    seed = 0x5eed * 1000003 + 7
    x = torch.ones(M, N, device=dev())
    y = torch.empty_like(x)
    L.dropout_apply(x, y, p, site, seed)
    ref = DropoutStream(seed).mask('rows', site, torch.empty(M, N), p)
    assert torch.equal(y.cpu(), ref)
    frac = float((ref == 0).float().mean())
    assert abs(frac - p) < 0.02, frac                      # and it IS a dropout of rate p


@pytest.mark.parametrize('S,dh,nh,causal', [(30, 64, 2, False), (20, 32, 2, False), (30, 16, 4, False), (50, 64, 2, False),
                                            (20, 32, 2, True), (50, 32, 2, True), (40, 64, 2, True), (100, 64, 1, True)])
def test_attention_masks_equal_the_librarys(S, dh, nh, causal):
    """probabilities are uniform for Q = K = 0; with V[k] = one-hot(k - off) the context row q holds keep(q, k) / S in column k - off:
    the mask as the kernel applied it, whatever its internal index (head dim < S: several windows).  causal (the user tower; above 32
    positions the key-masked causal form of the long kernels): row q is uniform over its q + 1 keys, the lower triangle is compared."""
    from adapter4rec_amd import _lib as L
    from oracle.dropout_masks import DropoutStream
    n_items, p, site, seed = 5, 0.1, 16, 0x5eed * 1000003 + 3
    H = nh * dh
    got = torch.zeros(n_items, nh, S, S)
    for off in range(0, S, dh):
        qkv = torch.zeros(n_items * S, 3 * H, device=dev())
        v = torch.zeros(n_items, S, nh, dh)
        for k in range(off, min(S, off + dh)):
            v[:, k, :, k - off] = 1.0
        qkv[:, 2 * H:] = v.reshape(n_items * S, H).to(dev())
        out = torch.zeros(n_items * S, H, device=dev())
        km = torch.ones(n_items, S, device=dev())
        if S <= 32:
            L.attn_fwd(qkv, out, km, n_items, S, nh, dh, 0, H, 2 * H, causal, 1.0, -1e9, drop_p=p, drop_site=site, drop_seed=seed)
        else:
            lse = torch.zeros(n_items * nh * S, device=dev())
            L.attn_long_fwd(qkv, out, lse, n_items, S, nh, dh, 0, H, 2 * H, 1.0, drop_p=p, drop_site=site, drop_seed=seed,
                            key_mask=km if causal else None, causal=causal)
        o = out.cpu().view(n_items, S, nh, dh).permute(0, 2, 1, 3)              # [item, head, q, column]
        w = min(S, off + dh) - off
        n_keys = torch.arange(1, S + 1, dtype=torch.float32)[None, None, :, None] if causal else float(S)
        got[:, :, :, off:off + w] = o[:, :, :, :w] * n_keys
    ref = DropoutStream(seed).mask('attn_item', site, torch.empty(n_items, nh, S, S), p, head_dim=dh)
    if causal:
        ref = torch.tril(ref)
    assert torch.allclose(got, ref, atol=1e-5), float((got - ref).abs().max())


def _oracle_cfg(cfg, args, geom, seed, eng=None):
    from oracle.dropout_masks import DropoutStream
    c = dict(cfg)
    c.update(drop=DropoutStream(seed, sasrec_fused=eng._sas_fused_ok() if eng is not None else True), p_hidden=float(geom.get('hidden_dropout_prob', 0.0)), p_attn=float(geom.get('attention_probs_dropout_prob', 0.0)),
             p_sas=float(args.drop_rate), drop_cls_only=bool(eng.cls_only) if eng is not None else True)
    return c


def _check_step(root, names, out, grads, loss, tol_loss, tol_grad):
    assert abs(loss.item() - float(out['loss'].detach())) < tol_loss, (loss.item(), float(out['loss'].detach()))
    params = dict(root.named_parameters())
    worst, where = 0.0, None
    for k in names:
        ref = grads[strip(k)].numpy()
        err = np.abs(params[k].grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-12)
        if err > worst:
            worst, where = err, k
    assert worst < tol_grad, (worst, where)
    return worst, where


@pytest.mark.parametrize('name', ['houlsby', 'pfeiffer', 'houlsby_parallel', 'compacter', 'roberta_cpc_pfeiffer'])
def test_step_fp32_dropout_on_vs_oracle_same_masks(name, monkeypatch):
    from oracle import ref_cpu as R
    import test_engine_gpu as T
    monkeypatch.setenv('A4R_SKIP_UNUSED_ITEMS', '0')       # every item slot encoded: the masks' row indices are the full batch's
    root, args, sd, cfg, fx, items, mask = T.build(name, 'fp32')
    root.train()
    inner = getattr(root, 'model', root)
    eng = inner._engine()
    eng.step_count = 0
    loss = root(items, mask, 0)
    loss.backward()
    seed = (eng.seed * 1000003 + eng.step_count) & 0xFFFFFFFFFFFF
    geom = dict(T.GEOM)
    ocfg = _oracle_cfg(cfg, args, geom, seed, eng)
    names = [str(k) for k in fx['trainable']]
    out, grads = R.loss_and_grads(sd, [strip(k) for k in names], items.cpu(), mask.cpu(), ocfg)
    assert abs(float(out['loss'].detach()) - float(fx['loss'])) > 1e-3          # the masks did change the step
    sites = {s for _, s in ocfg['drop'].used}
    assert {999, 0, 1, 2, 16, 17, 18, 4000, 4096, 4097, 4098, 4112, 4113, 4114} <= sites, sorted(sites)
    worst, where = _check_step(root, names, out, grads, loss, 1e-4, 1e-4)
    print(f'dropout ON, fp32 {name}: loss {loss.item():.6f} vs {float(out["loss"].detach()):.6f}, worst gradient {worst:.2e} of its max ({where})')


@pytest.mark.parametrize('name', ['houlsby', 'roberta_cpc_pfeiffer'])
def test_step_bf16_dropout_on_bound(name, monkeypatch):
    """the benched instantiation (bf16 storage, fp32 accumulate) in TRAINING mode against the fp32 oracle under the same masks: the bf16 bounds of
    test_step_bf16_bound (loss 2e-2, gradients 15 % of each tensor's max) hold with dropout on as well"""
    from oracle import ref_cpu as R
    import test_engine_gpu as T
    monkeypatch.setenv('A4R_SKIP_UNUSED_ITEMS', '0')
    root, args, sd, cfg, fx, items, mask = T.build(name, 'bf16')
    sd = T.condition(sd)
    root.load_state_dict({str(k): sd[strip(str(k))] for k in fx['all_keys']}, strict=True)
    root.train()
    inner = getattr(root, 'model', root)
    eng = inner._engine()
    eng.step_count = 0
    loss = root(items, mask, 0)
    loss.backward()
    seed = (eng.seed * 1000003 + eng.step_count) & 0xFFFFFFFFFFFF
    ocfg = _oracle_cfg(cfg, args, dict(T.GEOM), seed, eng)
    names = [str(k) for k in fx['trainable']]
    out, grads = R.loss_and_grads(sd, [strip(k) for k in names], items.cpu(), mask.cpu(), ocfg)
    worst, where = _check_step(root, names, out, grads, loss, 2e-2, 0.15)
    print(f'dropout ON, bf16 {name}: loss {loss.item():.5f} vs {float(out["loss"].detach()):.5f}, worst gradient {worst:.3f} of its max ({where})')


def test_step_fp32_dropout_on_vit_houlsby_same_masks(monkeypatch):
    """image tower (HF ViT: hidden / attention dropout 0.0 by configuration) + SASRec user tower with dropout on: the user tower's sites only"""
    from oracle import ref_cpu as R
    import test_engine_cv as TC
    monkeypatch.setenv('A4R_SKIP_UNUSED_ITEMS', '0')
    root, args, sd, cfg, fx, images, mask, noise = TC.build('cv_vit_houlsby', 'cuda:0', 'fp32')
    root.train()
    eng = getattr(root, 'model', root)._engine()
    eng.step_count = 0
    loss = root(images, mask, 0)
    loss.backward()
    seed = (eng.seed * 1000003 + eng.step_count) & 0xFFFFFFFFFFFF
    ocfg = _oracle_cfg(cfg, args, {}, seed, eng)
    names = [str(k) for k in fx['trainable']]
    out, grads = R.loss_and_grads(sd, [strip(k) for k in names], images.cpu(), mask.cpu(), ocfg)
    assert abs(float(out['loss'].detach()) - float(fx['loss'])) > 1e-3
    worst, where = _check_step(root, names, out, grads, loss, 1e-4, 1e-4)
    print(f'dropout ON, fp32 cv_vit_houlsby: loss {loss.item():.6f}, worst gradient {worst:.2e} ({where})')


@pytest.mark.parametrize('title,max_len', [(30, 40), (40, 20), (36, 33)])
def test_step_fp32_dropout_on_long_inputs_same_masks(title, max_len, monkeypatch):
    """dropout ON where the LONG attention kernels run: titles above 32 tokens (key mask) and histories above 32 positions (causal + key mask;
    parameters.py:29,44).  BERT-tiny + Houlsby, three users with histories of different lengths, fp32 instantiation vs the oracle under the same masks."""
    from adapter4rec_amd.inject import freeze_all, inject_adapters
    from adapter4rec_amd.model import BERT_BASE, BertBackbone, Model
    from base_cases import text_args
    from oracle import ref_cpu as R
    monkeypatch.setenv('A4R_SKIP_UNUSED_ITEMS', '0')
    torch.manual_seed(103)
    args = text_args('fp32', 'GELU')
    args.word_embedding_dim, args.bert_model_load, args.max_seq_len, args.num_words_title = 128, 'bert_tiny_uncased', max_len, title
    geom = dict(BERT_BASE, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512)
    model = Model(args, 512, True, BertBackbone(geom))
    freeze_all(model)
    model = inject_adapters(model, model.args)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.requires_grad:
                p.add_(0.02 * torch.randn_like(p))
    g = torch.Generator().manual_seed(10)
    Lq, users = max_len + 1, 3
    ids = torch.zeros(users, Lq, 2, 2 * title, dtype=torch.int64)
    mask = torch.zeros(users, Lq - 1)
    for u, n in enumerate((Lq, max(3, Lq // 2), 3)):
        for slot in range(Lq - n, Lq):
            for side in range(2):
                if side == 1 and slot == Lq - 1:
                    continue
                ln = int(torch.randint(4, title + 1, (1,), generator=g))
                ids[u, slot, side, 0] = 101
                ids[u, slot, side, 1:ln - 1] = torch.randint(1000, 30000, (ln - 2,), generator=g)
                ids[u, slot, side, ln - 1] = 102
                ids[u, slot, side, title:title + ln] = 1
        mask[u, Lq - n:] = 1
    items = ids.view(-1, 2 * title)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    cfg = dict(R.DEFAULT_CFG, adapter_activation='GELU', bert_heads=2, max_seq_len=max_len, num_words_title=title)
    out0, _ = R.loss_and_grads(sd, names, items, mask, cfg)
    model.to('cuda:0')
    model.train()
    eng = model._engine()
    eng.step_count = 0
    loss = model(items.to('cuda:0'), mask.to('cuda:0'), 0)
    loss.backward()
    seed = (eng.seed * 1000003 + eng.step_count) & 0xFFFFFFFFFFFF
    ocfg = _oracle_cfg(cfg, args, geom, seed, eng)
    out, grads = R.loss_and_grads(sd, names, items, mask, ocfg)
    assert abs(float(out['loss'].detach()) - float(out0['loss'].detach())) > 1e-3          # the masks did change the step
    worst, where = _check_step(model, names, out, grads, loss, 1e-4, 1e-4)
    print(f'dropout ON, fp32, {title}-token titles, {max_len} positions: loss {loss.item():.6f} vs {float(out["loss"].detach()):.6f}, worst gradient {worst:.2e} ({where})')
    model.cpu()
