"""Shapes the fixtures do not cover: other title lengths, history lengths, user counts, a user with the shortest possible
history, an all-PAD item inside the batch, 4 heads.  No reference fixture here: the oracle (pinned elsewhere) is the checker.
CPU: host logic through tests/sim_lib.py; GPU: the same cases through the C ABI."""
import argparse

import numpy as np
import pytest
import torch

import sim_lib

CASES = {
    'short_titles': dict(S=12, T=5, B=3, heads=2, layers=1),
    'one_user': dict(S=30, T=20, B=1, heads=2, layers=2),
    'long_history_32': dict(S=32, T=31, B=2, heads=4, layers=1),
    'five_users_cpc': dict(S=16, T=8, B=5, heads=2, layers=2, arch='cpc', adapter_type='pfeiffer', adapter_activation='relu'),
    # every adapter 64 wide (nothing zero-padded): the geometry whose gradient exchange is chunked and overlapped (tests/test_ddp_cpu.py)
    'four_users_wide_adapters': dict(S=16, T=8, B=4, heads=2, layers=3, adapter_down_size=64),
    'four_users_padded': dict(S=16, T=8, B=4, heads=2, layers=2),
}


def make(case, device, dtype='fp32'):
    from adapter4rec_amd.inject import freeze_all, inject_adapters
    from adapter4rec_amd.model import BertBackbone, Model, ModelCPC
    c = CASES[case]
    S, T, B = c['S'], c['T'], c['B']
    args = argparse.Namespace(
        max_seq_len=T, l2_weight=0, embedding_dim=64, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
        num_words_title=S, num_words_abstract=50, num_words_body=50, news_attributes=['title'], word_embedding_dim=128,
        bert_model_load='bert_tiny', bert_adapter_down_size=64, adapter_down_size=c.get('adapter_down_size', 16), adapter_dropout_rate=0.1,
        adapter_activation=c.get('adapter_activation', 'RELU'), hypercomplex_division=4, phm_init_range=1e-4,
        adapter_type=c.get('adapter_type', 'houslby'), is_serial='True', adding_adapter_to='all', arch=c.get('arch', 'sasrec'),
        compute_dtype=dtype)
    torch.manual_seed(sum(map(ord, case)))
    geom = dict(vocab_size=90, hidden_size=64 * c['heads'], num_hidden_layers=c['layers'], num_attention_heads=c['heads'],
                intermediate_size=128, max_position_embeddings=40, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
                attention_probs_dropout_prob=0.1, pad_token_id=0, model_type='bert')
    args.word_embedding_dim = geom['hidden_size']
    args.bert_model_load = {64: 'bert_tiny', 128: 'bert_tiny', 256: 'bert_mini'}[geom['hidden_size']] if geom['hidden_size'] != 64 else 'bert_tiny'
    model = (ModelCPC if args.arch == 'cpc' else Model)(args, 50, True, BertBackbone(geom))
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    freeze_all(model)
    model = inject_adapters(model, args)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.requires_grad:
                p.add_(0.05 * torch.randn_like(p))
    model.eval()
    L = T + 1
    g = torch.Generator().manual_seed(1)
    ids = torch.zeros(B, L, 2, 2 * S, dtype=torch.int64)
    mask = torch.zeros(B, T)
    for u in range(B):
        n = L if u % 2 == 0 else 2                       # full history / the shortest one (one input, one target)
        pad = L - n
        for slot in range(pad, L):
            for side in range(2):
                if side == 1 and slot == L - 1:
                    continue                              # last negative stays the PAD item
                ln = int(torch.randint(3, S + 1, (1,), generator=g))
                ids[u, slot, side, :ln] = torch.randint(1, 90, (ln,), generator=g)
                ids[u, slot, side, S:S + ln] = 1
        mask[u, pad:] = 1
    return model.to(device), args, geom, ids.view(-1, 2 * S).to(device), mask.to(device)


def check(case, device):
    from oracle import ref_cpu as R
    model, args, geom, items, mask = make(case, device)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cfg = dict(R.DEFAULT_CFG, bert_heads=geom['num_attention_heads'], max_seq_len=args.max_seq_len, num_words_title=args.num_words_title,
               arch=args.arch, adapter_type=args.adapter_type, adapter_activation=args.adapter_activation)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    out, grads = R.loss_and_grads(sd, names, items.cpu(), mask.cpu(), cfg)
    loss = model(items, mask, device)
    loss.backward()
    assert abs(loss.item() - float(out['loss'].detach())) < 1e-4 * max(1.0, abs(float(out['loss'].detach())))
    np.testing.assert_allclose(model.bert_encoder(items).cpu().numpy(), out['input_embs_all'].detach().numpy(), atol=1e-4, rtol=0)
    params = dict(model.named_parameters())
    for n in names:
        ref = grads[n].numpy()
        np.testing.assert_allclose(params[n].grad.cpu().numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=n)


@pytest.fixture
def simulated(monkeypatch):
    import adapter4rec_amd.engine as E
    import adapter4rec_amd.optim as O
    monkeypatch.setattr(E, 'L', sim_lib)
    monkeypatch.setattr(O, 'L', sim_lib)
    monkeypatch.setattr(E.TransRecEngine, '_require_device', lambda self, p0: None)


@pytest.mark.parametrize('case', list(CASES))
def test_shapes_host_logic(simulated, case):
    check(case, 'cpu')


@pytest.mark.gpu
@pytest.mark.parametrize('case', list(CASES))
def test_shapes_gpu(case):
    check(case, 'cuda:0')


def large_then_small(case, device):
    """A smaller batch after a larger one on the SAME engine (the last batch of every epoch: run.py's DataLoader has no
    drop_last, Downstream/Text/run.py:356) must give the gradients of a fresh engine: the cached work buffers are sliced to a
    padded row count and the padding rows of the gradient buffers have to be exact zeros again."""
    c = CASES[case]
    L2 = 2 * (c['T'] + 1)

    def grads(model, items, mask):
        model.zero_grad()
        loss = model(items, mask, device)
        loss.backward()
        return loss.item(), {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}

    model, args, geom, items, mask = make(case, device)
    grads(model, items, mask)                                   # B users
    l_small, g_small = grads(model, items[:L2], mask[:1])       # then 1 user on the same buffers
    fresh, *_ = make(case, device)
    l_ref, g_ref = grads(fresh, items[:L2], mask[:1])
    assert abs(l_small - l_ref) < 1e-6 * max(1.0, abs(l_ref))
    for n, ref in g_ref.items():
        np.testing.assert_allclose(g_small[n].numpy(), ref.numpy(), atol=1e-7 + 2e-6 * ref.abs().max().item(), rtol=0, err_msg=n)


@pytest.mark.parametrize('case', ['short_titles', 'five_users_cpc'])
def test_large_then_small_batch_host_logic(simulated, case):
    large_then_small(case, 'cpu')


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['short_titles', 'five_users_cpc', 'long_history_32'])
def test_large_then_small_batch_gpu(case):
    large_then_small(case, 'cuda:0')


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['one_user', 'four_users_wide_adapters'])
def test_wgrad_side_stream_gpu(monkeypatch, case):
    """A4R_WGRAD_STREAM=1 (adapter weight gradients on a side stream; off by default since the end of round 2) against the oracle."""
    import adapter4rec_amd.engine as E
    monkeypatch.setattr(E, 'WGRAD_STREAM', True)
    check(case, 'cuda:0')
