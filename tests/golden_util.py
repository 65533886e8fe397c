"""Load tests/golden fixtures (written by tools/gen_golden.py from the imported reference)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

VARIANT_CFG = {
    'houlsby': dict(),
    'houlsby_gelu': dict(adapter_activation='GELU'),
    'houlsby_parallel': dict(is_serial='None'),
    'pfeiffer': dict(adapter_type='pfeiffer', adapter_activation='relu'),
    'pfeiffer_ver2': dict(adapter_type='pfeiffer_ver2'),
    'compacter': dict(adapter_type='compacter'),
    'houlsby_cpc': dict(arch='cpc'),
    'finetune_all': dict(adapter_type='none'),
    'prompt': dict(adapter_type='prompt', n_tokens=8),
    'kadapter': dict(adapter_type='kadapter', k_adapter_bert_list='0,1', num_adapter_heads_bert=4, num_adapter_heads_sasrec=2),
    'roberta_cpc_pfeiffer': dict(adapter_type='pfeiffer', adapter_activation='relu', arch='cpc',
                                 encoder='roberta', bert_ln_eps=1e-5, pad_token_id=1),
    'roberta_prompt': dict(adapter_type='prompt', n_tokens=8, arch='cpc', encoder='roberta', bert_ln_eps=1e-5, pad_token_id=1),
}
LRS = dict(fine_tune_lr=5e-5, lr=1e-4, adapter_bert_lr=1.5e-4, adapter_sasrec_lr=1.5e-4)


def base_name(k):
    if k.startswith('model.'):
        k = k[len('model.'):]
    return k.replace('.self_output.', '.').replace('.transformer_block.', '.').replace('.word_embeddings.wte.', '.word_embeddings.') \
        .replace('.bert_model.bert_model.', '.bert_model.').replace('.transformer_blocks.transformer_blocks.', '.transformer_blocks.') \
        .replace('.encoder.vit_encoder.', '.encoder.')


def strip(k):
    return k[len('model.'):] if k.startswith('model.') else k


def load_variant(name):
    """-> (sd with reference key names ('model.' prefix of CompacterModel stripped), cfg, fixture dict, trainable names)."""
    from oracle.ref_cpu import DEFAULT_CFG
    base_file = 'base_roberta.npz' if name.startswith('roberta') else 'base.npz'
    base = np.load(os.path.join(GOLDEN, base_file))
    fx = np.load(os.path.join(GOLDEN, name + '.npz'))
    base_sd = {k[3:]: torch.from_numpy(base[k]) for k in base.files if k.startswith('sd/')}
    sd = {}
    for k in fx['all_keys']:
        k = str(k)
        if 'sd/' + k in fx.files:
            sd[strip(k)] = torch.from_numpy(fx['sd/' + k])
        else:
            sd[strip(k)] = base_sd[base_name(k)]
    cfg = dict(DEFAULT_CFG)
    cfg.update(bert_heads=2)
    cfg.update(VARIANT_CFG[name])
    trainable = [strip(str(k)) for k in fx['trainable']]
    batch = (torch.from_numpy(base['sample_items']).view(-1, 60), torch.from_numpy(base['log_mask']))
    return sd, cfg, fx, trainable, batch, base


def load_multi_attr():
    """multi_attr.npz (tools/gen_golden_r5.py: the reference with --news_attributes title,abstract on the base.npz weights + Houlsby adapters)
    -> (sd, cfg, fixture, trainable names, (sample_items [B * L * 2, 2 * (30 + 36)], log_mask))"""
    from oracle.ref_cpu import DEFAULT_CFG
    base = np.load(os.path.join(GOLDEN, 'base.npz'))
    fx = np.load(os.path.join(GOLDEN, 'multi_attr.npz'))
    base_sd = {k[3:]: torch.from_numpy(base[k]) for k in base.files if k.startswith('sd/')}
    sd = {}
    for k in fx['all_keys']:
        k = str(k)
        sd[strip(k)] = torch.from_numpy(fx['sd/' + k]) if 'sd/' + k in fx.files else base_sd[base_name(k)]
    nt, na = (int(x) for x in fx['num_words'])
    cfg = dict(DEFAULT_CFG, bert_heads=2, news_attributes=['title', 'abstract'], num_words_title=nt, num_words_abstract=na)
    return sd, cfg, fx, [strip(str(k)) for k in fx['trainable']], (torch.from_numpy(fx['sample_items']), torch.from_numpy(fx['log_mask']))


# ---------------------------------------------------------------- image path fixtures (tools/gen_golden_cv.py)
CV_VARIANT_CFG = {
    'cv_vit_houlsby': dict(),
    'cv_vit_houlsby_gelu_ln': dict(adapter_activation='GELU'),
    'cv_vit_pfeiffer_ver2': dict(adapter_type='pfeiffer_ver2'),
    'cv_vit_compacter': dict(adapter_type='compacter'),
    'cv_vit_cpc': dict(arch='cpc'),
    'cv_vit_parallel': dict(is_serial='None'),
    'cv_vit_prompt': dict(adapter_type='prompt'),
    'cv_vit_kadapter': dict(adapter_type='kadapter', k_adapter_bert_list='0,1', num_adapter_heads_bert=2, num_adapter_heads_sasrec=2),
    'cv_mae_houlsby': dict(mae=True),
    'cv_vit_frozen': dict(adapter_type='none'),
}
CV_LRS = dict(fine_tune_lr=1e-5, lr=1e-3, adapter_cv_lr=5e-4, adapter_sasrec_lr=1e-4)


def load_cv_variant(name):
    """-> (sd, cfg, fixture, trainable names, (images [n,3,R,R], log_mask), noise)."""
    from oracle.ref_cpu import DEFAULT_CFG
    common = np.load(os.path.join(GOLDEN, 'cv_base.npz'))
    base = np.load(os.path.join(GOLDEN, 'cv_base_mae.npz')) if 'mae' in name else common
    fx = np.load(os.path.join(GOLDEN, name + '.npz'))
    base_sd = {k[3:]: torch.from_numpy(base[k]) for k in base.files if k.startswith('sd/')}
    sd = {}
    for k in fx['all_keys']:
        k = str(k)
        sd[strip(k)] = torch.from_numpy(fx['sd/' + k]) if 'sd/' + k in fx.files else base_sd[base_name(k)]
    cfg = dict(DEFAULT_CFG)
    cfg.update(tower='image', vit_heads=2, noise=torch.from_numpy(common['noise']))
    cfg.update(CV_VARIANT_CFG[name])
    trainable = [strip(str(k)) for k in fx['trainable']]
    batch = (torch.from_numpy(common['images']), torch.from_numpy(common['log_mask']))
    return sd, cfg, fx, trainable, batch, cfg['noise']


# ---------------------------------------------------------------- LoRA pinned through merged weights (tools/gen_golden_r4.py)
def lora_pin_case(tower, r_enc=8, r_sas=4, seed=11):
    """loralib is absent, but W x + b + (B A / r) x is the reference's own plain Linear at the merged weight W + B A / r.  With
    W := W_base - B A / r the LoRA layer IS the pinned base layer: its forward must equal the imported reference's numbers in
    lora_pin_<tower>.npz and its gradients follow from the reference's dL/dW by the chain rule
        dA = B^T dW / r,   dB = dW A^T / r,   dbias = the reference's dbias.
    -> (sd with lora_A / lora_B / shifted weight, oracle cfg, batch, pin fixture, {name: expected gradient}).
    Image tower: SASRec w_V is a plain trainable Linear (Downstream/CV/run_adapter.py:394, r = 0): its expected gradient is dW itself."""
    pin = np.load(os.path.join(GOLDEN, 'lora_pin_%s.npz' % tower))
    if tower == 'text':
        sd, cfg, _, _, batch, _ = load_variant('finetune_all')
        cfg = dict(cfg, adapter_type='lora', lora_r_bert=r_enc, lora_r_sasrec=r_sas)
    else:
        sd, cfg, _, _, batch, _ = load_cv_variant('cv_vit_frozen')
        cfg = dict(cfg, adapter_type='lora', lora_r_vit=r_enc, lora_r_sasrec=r_sas)
    sd = dict(sd)
    g = torch.Generator().manual_seed(seed)
    expect = {}
    for k in pin.files:
        if not k.startswith('grad/'):
            continue
        n = k[5:]
        dW = torch.from_numpy(pin[k])
        if n.endswith('.bias'):
            expect[n] = dW
            continue
        p = n[:-len('weight')]
        sas = 'multi_head_attention' in p
        if tower == 'image' and p.endswith('w_V.'):
            expect[n] = dW
            continue
        r = r_sas if sas else r_enc
        W = sd[n]
        A = torch.randn(r, W.shape[1], generator=g) * 0.3
        B = torch.randn(W.shape[0], r, generator=g) * 0.3
        sd[p + 'lora_A'], sd[p + 'lora_B'] = A, B
        sd[n] = W - (B @ A) / r
        expect[p + 'lora_A'] = (B.t() @ dW) / r
        expect[p + 'lora_B'] = (dW @ A.t()) / r
    return sd, cfg, batch, pin, expect


# ---- round 6: the reference's readers on its own shipped data files (tools/gen_golden_r6.py, tests/test_real_data.py) ----
def sha_array(a):
    """SHA-256 of an array's dtype, shape and C-order bytes."""
    import hashlib
    a = np.ascontiguousarray(a)
    h = hashlib.sha256()
    h.update(f'{a.dtype.str}|{a.shape}|'.encode())
    h.update(a.tobytes())
    return h.hexdigest()


def sha_mapping(d):
    """SHA-256 of a dict in ITERATION order (the readers' dicts are insertion-ordered and that order is the user / item numbering):
    one line per entry, 'key<TAB>value'; lists / tensors / arrays as space-separated integers, bytes decoded."""
    import hashlib
    h = hashlib.sha256()
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            v = v.tolist()
        elif isinstance(v, np.ndarray):
            v = v.tolist()
        if isinstance(v, (list, tuple)):
            v = ' '.join(str(int(x)) for x in v)
        elif isinstance(v, bytes):
            v = v.decode('ascii')
        k = k.decode('ascii') if isinstance(k, bytes) else k
        h.update(f'{k}\t{v}\n'.encode())
    return h.hexdigest()
