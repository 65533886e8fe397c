#!/usr/bin/env python
"""Generate tests/golden/cv_*.npz by IMPORTING the reference's image path (CPU, build container only).

What is the reference's own code here: ``Model`` / ``ModelCPC`` / ``Vit_Encoder`` / ``MAE_Encoder`` / ``User_Encoder``, the
wrappers ``VITAdaptedSelfOutput``, ``VITAdaptedOutput``, ``VITCompacterAdapted*``, ``SASRec*AdaptedSelfOutput`` and the
adapter blocks (Downstream/CV/model).  What is NOT: the HuggingFace backbone.  The reference addresses transformers==4.20.1's
module tree (``vit.encoder.layer[i].attention.attention.query`` ...), which the installed transformers no longer has, so the
backbone below is a 4.20.1-shaped re-statement (third party: modeling_vit.py / modeling_vit_mae.py) that this script first
checks against the INSTALLED HuggingFace ViT / ViT-MAE forward with mapped weights (max |diff| stored in the fixture).
The wrappers hard-code width 768; for the tiny geometry their ``adapter`` attribute is re-created at the tiny width with the
reference's own AdapterBlock class (placement logic = the wrappers' forward stays the reference's).
"""
import argparse
import copy
import os
import sys

import numpy as np
import torch
from torch import nn

REF = '/root/reference/Downstream/CV'
sys.path.insert(0, REF)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')

import model as refm  # noqa: E402
from model import Model, ModelCPC  # noqa: E402
from model.model import VITKAdaptedCVModel, SASRecKAdaptedTransformerBlocks  # noqa: E402
from model.modules import KAdapterBlock  # noqa: E402
from model.model import (SoftPrompt, VITAdaptedParallelOutput, SASRecParallelAdaptedSelfOutput, VITAdaptedSelfOutput, VITAdaptedOutput, VITCompacterAdaptedSelfOutput, VITCompacterAdaptedOutput,  # noqa: E402
                         SASRecAdaptedSelfOutput, SASRecPfeifferV2AdaptedSelfOutput, SASRecCompacterAdaptedSelfOutput)
from model.modules import AdapterBlock, HyperComplexAdapterBlock  # noqa: E402
from model.layers import PHMLinear  # noqa: E402

HID, LAYERS, HEADS, FFN, IMG, PATCH = 128, 2, 2, 256, 32, 8
ITEM_NUM, B, L, E = 60, 2, 21, 64
LRS = dict(fine_tune_lr=1e-5, lr=1e-3, adapter_cv_lr=5e-4, adapter_sasrec_lr=1e-4)      # Downstream/CV/parameters.py defaults


def make_args(**kw):
    a = argparse.Namespace(max_seq_len=20, l2_weight=0, embedding_dim=E, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
                           CV_model_load='vit-base-patch16-224', cv_adapter_down_size=64, adapter_down_size=16, adapter_dropout_rate=0.1,
                           adapter_activation='RELU', hypercomplex_division=4, phm_init_range=1e-4, adapter_type='houslby',
                           is_serial='True', arch='sasrec')
    for k, v in kw.items():
        setattr(a, k, v)
    return a


# ------------------------------------------------------------------ transformers==4.20.1-shaped backbone (third party, restated)
class SelfAttention(nn.Module):
    def __init__(self):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(HID, HID), nn.Linear(HID, HID), nn.Linear(HID, HID)
        self.dropout = nn.Dropout(0.0)

    def forward(self, x):
        n, s, _ = x.shape
        sp = lambda t: t.view(n, s, HEADS, HID // HEADS).transpose(1, 2)
        p = torch.softmax(sp(self.query(x)) @ sp(self.key(x)).transpose(-1, -2) / (HID // HEADS) ** 0.5, -1)
        return (self.dropout(p) @ sp(self.value(x))).transpose(1, 2).reshape(n, s, HID)


class SelfOutput(nn.Module):          # ViTSelfOutput: the residual is added in ViTLayer
    def __init__(self, inp):
        super().__init__()
        self.dense, self.dropout = nn.Linear(inp, HID), nn.Dropout(0.0)

    def forward(self, hidden_states, input_tensor):
        return self.dropout(self.dense(hidden_states))


class Output(SelfOutput):             # ViTOutput
    def forward(self, hidden_states, input_tensor):
        return self.dropout(self.dense(hidden_states)) + input_tensor


class Attention(nn.Module):
    def __init__(self):
        super().__init__()
        self.attention, self.output = SelfAttention(), SelfOutput(HID)

    def forward(self, x):
        return self.output(self.attention(x), x)


class Intermediate(nn.Module):
    def __init__(self):
        super().__init__()
        self.dense = nn.Linear(HID, FFN)

    def forward(self, x):
        return torch.nn.functional.gelu(self.dense(x))


class Layer(nn.Module):
    def __init__(self):
        super().__init__()
        self.attention, self.intermediate, self.output = Attention(), Intermediate(), Output(FFN)
        self.layernorm_before, self.layernorm_after = nn.LayerNorm(HID, eps=1e-12), nn.LayerNorm(HID, eps=1e-12)

    def forward(self, x):
        x = self.attention(self.layernorm_before(x)) + x
        return self.output(self.intermediate(self.layernorm_after(x)), x)


class Encoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.layer = nn.ModuleList([Layer() for _ in range(LAYERS)])

    def forward(self, x, head_mask=None, output_attentions=False, output_hidden_states=False, return_dict=True):
        """4.20.1 ViTEncoder.forward's signature; with output_hidden_states the tuple form (last, all hidden states = the embedding
        output followed by every layer's output, attentions [not produced here]) that VITKAdaptedCVModel indexes (model.py:388-391)."""
        hs = (x,)
        for l_ in self.layer:
            x = l_(x)
            hs = hs + (x,)
        return (x, hs, ()) if output_hidden_states else x


class PatchEmbeddings(nn.Module):
    def __init__(self):
        super().__init__()
        self.projection = nn.Conv2d(3, HID, kernel_size=PATCH, stride=PATCH)

    def forward(self, px):
        return self.projection(px).flatten(2).transpose(1, 2)


class Embeddings(nn.Module):
    def __init__(self, mae):
        super().__init__()
        n = (IMG // PATCH) ** 2
        self.mae = mae
        self.cls_token = nn.Parameter(torch.randn(1, 1, HID) * 0.02)
        self.patch_embeddings = PatchEmbeddings()
        self.position_embeddings = nn.Parameter(torch.randn(1, n + 1, HID) * 0.02, requires_grad=not mae)
        self.dropout = nn.Dropout(0.0)
        self.noise = None

    def forward(self, px):
        x = self.patch_embeddings(px)
        n = x.shape[0]
        if not self.mae:
            return self.dropout(torch.cat([self.cls_token.expand(n, -1, -1), x], 1) + self.position_embeddings)
        x = x + self.position_embeddings[:, 1:]
        keep = torch.argsort(self.noise, dim=1)[:, :int(x.shape[1] * 0.25)]
        x = torch.gather(x, 1, keep[:, :, None].expand(-1, -1, HID))
        return torch.cat([(self.cls_token + self.position_embeddings[:, :1]).expand(n, -1, -1), x], 1)


class ViTModel(nn.Module):
    def __init__(self, mae=False):
        super().__init__()
        self.embeddings, self.encoder, self.layernorm = Embeddings(mae), Encoder(), nn.LayerNorm(HID, eps=1e-12)

    def forward(self, px, **kw):        # 4.20.1 ViTModel.forward: keyword call into the encoder, sequence output = element 0
        enc = self.encoder(self.embeddings(px), head_mask=None, output_attentions=None, output_hidden_states=None, return_dict=None)
        return (self.layernorm(enc if torch.is_tensor(enc) else enc[0]),)


class ViTForImageClassification(nn.Module):
    def __init__(self):
        super().__init__()
        self.vit, self.classifier = ViTModel(), nn.Linear(HID, E)

    def forward(self, px, return_dict=None):
        return (self.classifier(self.vit(px)[0][:, 0]),)


def check_against_installed_hf(vit_cls, mae_model, images, noise):
    """The restated backbone == the installed HuggingFace ViT / ViT-MAE (eager attention) on mapped weights."""
    from transformers import ViTConfig, ViTMAEConfig
    from transformers import ViTForImageClassification as HFViT, ViTMAEModel as HFMAE
    kw = dict(hidden_size=HID, num_hidden_layers=LAYERS, num_attention_heads=HEADS, intermediate_size=FFN, image_size=IMG,
              patch_size=PATCH, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=1e-12)

    def remap(sd, pre):
        out = {}
        for k, v in sd.items():
            k2 = (k.replace('encoder.layer.', 'layers.').replace('attention.attention.query', 'attention.q_proj')
                  .replace('attention.attention.key', 'attention.k_proj').replace('attention.attention.value', 'attention.v_proj')
                  .replace('attention.output.dense', 'attention.o_proj').replace('intermediate.dense', 'mlp.fc1')
                  .replace('output.dense', 'mlp.fc2'))
            out[k2] = v
        return out
    c = ViTConfig(num_labels=E, **kw)
    c._attn_implementation = 'eager'
    hf = HFViT(c).eval()
    missing = hf.load_state_dict(remap(vit_cls.state_dict(), ''), strict=True)
    with torch.no_grad():
        d1 = float((hf(images).logits - vit_cls(images)[0]).abs().max())
    c2 = ViTMAEConfig(mask_ratio=0.75, **kw)
    c2._attn_implementation = 'eager'
    hm = HFMAE(c2).eval()
    hm.load_state_dict(remap(mae_model.state_dict(), ''), strict=True)
    mae_model.embeddings.noise = noise
    with torch.no_grad():
        d2 = float((hm(images, noise=noise).last_hidden_state - mae_model(images)[0]).abs().max())
    print(f'restated 4.20.1 backbone vs installed HF: ViT max|diff| {d1:.2e}, ViT-MAE {d2:.2e}')
    assert d1 < 2e-5 and d2 < 2e-5
    return d1, d2


class CompacterModel(nn.Module):       # Downstream/CV/run_adapter.py:85-99
    def __init__(self, args, model):
        super().__init__()
        n = args.hypercomplex_division
        self.model = model
        self.phm_rule = nn.Parameter(torch.FloatTensor(n, n, n))
        self.phm_rule.data.normal_(mean=0, std=args.phm_init_range)
        for _, sub in self.model.named_modules():
            if isinstance(sub, PHMLinear):
                sub.set_phm_rule(phm_rule=self.phm_rule)

    def forward(self, sample_items, log_mask, local_rank):
        return self.model(sample_items, log_mask, local_rank)


def layers_of(m):
    net = m.cv_encoder.image_net
    return net.vit.encoder.layer if hasattr(net, 'vit') else net.encoder.layer


def inject_kadapter(m, args):          # Downstream/CV/run_adapter.py:378-383 at the tiny width
    net = m.cv_encoder.image_net
    w = VITKAdaptedCVModel(net.vit.encoder, args)                 # hard-codes 768: the adapter list / com_dense are re-created at HID
    w.bert_adapter_list = nn.ModuleList([KAdapterBlock(args, args.num_adapter_heads_bert, HID, args.k_adapter_bert_hidden_dim,
                                                       args.adapter_dropout_rate) for _ in w.k_adapter_num_list])
    w.com_dense = nn.Linear(HID * 2, HID)
    net.vit.encoder = w
    te = m.user_encoder.transformer_encoder
    te.transformer_blocks = SASRecKAdaptedTransformerBlocks(te.transformer_blocks, args)
    return m


def inject(m, args):                   # Downstream/CV/run_adapter.py:369-447 at the tiny width (see module docstring)
    t = args.adapter_type
    blocks = m.user_encoder.transformer_encoder.transformer_blocks
    def wrap(cls, so, block_cls):
        w = cls(so, args)
        w.adapter = block_cls(args, HID, args.cv_adapter_down_size) if block_cls is HyperComplexAdapterBlock else \
            block_cls(args, HID, args.cv_adapter_down_size, args.adapter_dropout_rate)
        return w
    if 'prompt' in t:                       # run_adapter.py:413-422
        net = m.cv_encoder.image_net
        net.vit.embeddings = SoftPrompt(net.vit.embeddings, n_tokens=args.n_tokens, embed_dim=HID)
        for n_, p in m.named_parameters():
            if 'cv_encoder.image_net.classifier' in n_:
                p.requires_grad = True
        return m
    if 'kadapter' in t:
        return inject_kadapter(m, args)
    if 'pfeiffer_ver2' in t:
        for lyr in layers_of(m):
            lyr.attention.output = wrap(VITAdaptedSelfOutput, lyr.attention.output, AdapterBlock)
        for i, blk in enumerate(blocks):
            blocks[i] = SASRecPfeifferV2AdaptedSelfOutput(blk, args)
    elif 'compacter' in t:
        for lyr in layers_of(m):
            lyr.attention.output = wrap(VITCompacterAdaptedSelfOutput, lyr.attention.output, HyperComplexAdapterBlock)
            lyr.output = wrap(VITCompacterAdaptedOutput, lyr.output, HyperComplexAdapterBlock)
        for i, blk in enumerate(blocks):
            blocks[i] = SASRecCompacterAdaptedSelfOutput(blk, args)
        m = CompacterModel(args, m)
    elif 'houslby' in t and 'None' in args.is_serial:      # run_adapter.py:448-460
        for lyr in layers_of(m):
            lyr.output = wrap(VITAdaptedParallelOutput, lyr.output, AdapterBlock)
        for i, blk in enumerate(blocks):
            blocks[i] = SASRecParallelAdaptedSelfOutput(blk, args)
    elif 'houslby' in t:
        for lyr in layers_of(m):
            lyr.attention.output = wrap(VITAdaptedSelfOutput, lyr.attention.output, AdapterBlock)
            lyr.output = wrap(VITAdaptedOutput, lyr.output, AdapterBlock)
        for i, blk in enumerate(blocks):
            blocks[i] = SASRecAdaptedSelfOutput(blk, args)
    return m


def optimizer_for(m):                  # run_adapter.py:491-517
    g = dict(img=[], rec=[], acv=[], arec=[])
    for name, p in m.named_parameters():
        if not p.requires_grad:
            continue
        ad = 'adapter' in name
        if 'image_net' in name and not ('fc' in name or 'classifier' in name or 'decoder_pred' in name):
            g['acv' if ad else 'img'].append(p)
        else:
            g['arec' if ad else 'rec'].append(p)
    return torch.optim.Adam([{'params': g['img'], 'lr': LRS['fine_tune_lr']}, {'params': g['rec'], 'lr': LRS['lr']},
                             {'params': g['acv'], 'lr': LRS['adapter_cv_lr']}, {'params': g['arec'], 'lr': LRS['adapter_sasrec_lr']}])


def base_name(k):
    if k.startswith('model.'):
        k = k[len('model.'):]
    return k.replace('.self_output.', '.').replace('.transformer_block.', '.').replace('.encoder.vit_encoder.', '.encoder.') \
        .replace('.transformer_blocks.transformer_blocks.', '.transformer_blocks.')


def run_variant(name, base_model, images, masks, noise, args, layernorm=False, autocast_only=False):
    torch.manual_seed(2000 + sum(map(ord, name)))
    m = copy.deepcopy(base_model)
    if args.arch == 'cpc':
        c = ModelCPC(args, ITEM_NUM, True, m.cv_encoder.image_net)
        c.cv_encoder, c.user_encoder = m.cv_encoder, m.user_encoder
        m = c
    for p in m.parameters():
        p.requires_grad = False
    m = inject(m, args)
    if layernorm:                       # --finetune_layernorm (run_adapter.py:484-488)
        for n_, p in m.named_parameters():
            if 'adapter' not in n_ and ('LayerNorm' in n_ or 'layer_norm' in n_ or 'layernorm' in n_):
                p.requires_grad = True
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if p.requires_grad and ('adapter' in n_ or n_.endswith('phm_rule') or 'Prompt_Tokens' in n_):
                p.add_(0.05 * torch.randn_like(p))
    m.eval()
    inner = m.model if isinstance(m, CompacterModel) else m
    emb_mod = inner.cv_encoder.image_net.embeddings if 'mae' in args.CV_model_load else inner.cv_encoder.image_net.vit.embeddings
    getattr(emb_mod, 'wte', emb_mod).noise = noise
    out = {}
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    base_sd = base_model.state_dict()
    out['all_keys'] = np.array(list(sd.keys()))
    for k, v in sd.items():
        bk = base_name(k)
        if not (bk in base_sd and torch.equal(base_sd[bk], v)):
            out['sd/' + k] = v.numpy()
    trainable = [n_ for n_, p in m.named_parameters() if p.requires_grad]
    out['trainable'] = np.array(trainable)
    with torch.no_grad():
        embs = inner.cv_encoder(images)
        e = embs.view(-1, L, 2, E)
        prec = inner.user_encoder(e[:, :-1, 0], masks, 'cpu')
    out['input_embs_all'] = embs.numpy()
    out['prec_vec'] = prec.numpy()
    m.zero_grad()
    loss = m(images, masks, 'cpu')
    out['loss'] = loss.detach().numpy()
    if trainable:
        loss.backward()
    for n_, p in m.named_parameters():
        if p.requires_grad:
            out['grad/' + n_] = p.grad.detach().numpy().copy()
    if autocast_only:
        # round 6 (VERDICT r5 housekeeping): the same weights and batch once more under torch.autocast(bfloat16) -- the reference's reduced-precision
        # path (`with autocast(): bz_loss = model(...)`, Downstream/CV/run_adapter.py:565-593; fp16 + GradScaler on CUDA, bf16 here as for the text
        # tower's *_autocast.npz) -> <name>_autocast.npz: loss, embeddings, every trainable gradient; the fp32 numbers stay in <name>.npz
        fx = np.load(os.path.join(OUT, name + '.npz'))
        assert abs(float(fx['loss']) - float(loss)) < 1e-6, 'the rebuilt model is not the fixture\'s'
        m.zero_grad()
        with torch.autocast('cpu', dtype=torch.bfloat16):
            l_ac = m(images, masks, 'cpu')
            embs_ac = inner.cv_encoder(images)
        l_ac.float().backward()
        ac = dict(loss=np.array(float(l_ac.detach().float())), input_embs_all=embs_ac.detach().float().numpy())
        for n_, p in m.named_parameters():
            if p.requires_grad:
                ac['grad/' + n_] = p.grad.detach().float().numpy().copy()
        np.savez_compressed(os.path.join(OUT, name + '_autocast.npz'), **ac)
        print(f"{name}_autocast: loss {float(ac['loss']):.6f} (fp32 {float(loss):.6f}); emb err {np.abs(ac['input_embs_all'] - out['input_embs_all']).max():.3e}")
        return
    if trainable:
        opt = optimizer_for(m)
        losses = []
        for s in range(3):
            opt.zero_grad()
            l_ = m(images, masks, 'cpu')
            l_.backward()
            opt.step()
            losses.append(float(l_.detach()))
            if s in (0, 2):
                for n_, p in m.named_parameters():
                    if p.requires_grad:
                        out[f'adam{s + 1}/' + n_] = p.detach().numpy().copy()
        out['adam_losses'] = np.array(losses)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(f'{name}: loss {float(loss):.6f}  trainable {len(trainable)} tensors')


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(123456)
    rng = np.random.default_rng(123456)
    # batch: 2 users (full history, short history), slots [L, 2]; pad slots are all-zero images (dataset.py:91,112)
    u8 = rng.integers(0, 256, size=(B, L, 2, IMG, IMG, 3), dtype=np.uint8)
    images = ((torch.from_numpy(u8).float() / 255 - 0.5) / 0.5).permute(0, 1, 2, 5, 3, 4).contiguous()
    lens = [21, 9]
    masks = torch.zeros(B, L - 1)
    for u, n in enumerate(lens):
        pad = L - n
        images[u, :pad] = 0
        images[u, :, 1][-1] = 0                         # last negative slot is never filled (dataset.py:94-105)
        masks[u, pad:] = 1
    images = images.view(-1, 3, IMG, IMG)
    noise = torch.from_numpy(rng.random((images.shape[0], (IMG // PATCH) ** 2))).float()

    args = make_args()
    vit = ViTForImageClassification()
    nn.init.xavier_normal_(vit.classifier.weight)      # run_adapter.py:293-296
    nn.init.zeros_(vit.classifier.bias)
    for p_ in vit.vit.parameters():                    # spread the LayerNorm / bias tensors away from (1, 0)
        if p_.dim() == 1:
            p_.data.add_(0.1 * torch.randn_like(p_))
    base = Model(args, ITEM_NUM, True, vit).eval()
    mae_args = make_args(CV_model_load='vit-mae-base')
    mae_net = ViTModel(mae=True)
    for p_ in mae_net.parameters():
        if p_.dim() == 1:
            p_.data.add_(0.1 * torch.randn_like(p_))
    base_mae = Model(mae_args, ITEM_NUM, True, mae_net)
    base_mae.cv_encoder.cv_proj = nn.Linear(HID, E)     # MAE_Encoder hard-codes 768 -> 64 (encoders.py:12-15); tiny width here
    nn.init.xavier_normal_(base_mae.cv_encoder.cv_proj.weight)
    nn.init.zeros_(base_mae.cv_encoder.cv_proj.bias)
    base_mae.user_encoder.load_state_dict(base.user_encoder.state_dict())
    base_mae.eval()
    d1, d2 = check_against_installed_hf(vit, mae_net, images[:6], noise[:6])

    if not (len(sys.argv) > 1 and sys.argv[1] in ('--parallel-only', '--prompt-only', '--kadapter-only', '--autocast-only')):
      np.savez_compressed(os.path.join(OUT, 'cv_base.npz'), images=images.numpy(), log_mask=masks.numpy(), noise=noise.numpy(),
                        hf_check=np.array([d1, d2]), **{'sd/' + k: v.numpy() for k, v in base.state_dict().items()})
      np.savez_compressed(os.path.join(OUT, 'cv_base_mae.npz'), **{'sd/' + k: v.numpy() for k, v in base_mae.state_dict().items()})
    if len(sys.argv) > 1 and sys.argv[1] == '--parallel-only':      # added later: leaves the other fixtures untouched
        run_variant('cv_vit_parallel', base, images, masks, noise, make_args(is_serial='None'))
        return
    if len(sys.argv) > 1 and sys.argv[1] == '--kadapter-only':     # added in round 2
        run_variant('cv_vit_kadapter', base, images, masks, noise,
                    make_args(adapter_type='kadapter', k_adapter_bert_list='0,1', k_adapter_bert_hidden_dim=64, num_adapter_heads_bert=2,
                              num_adapter_heads_sasrec=2))
        return
    if len(sys.argv) > 1 and sys.argv[1] == '--autocast-only':     # added in round 6: <name>_autocast.npz for two variants, the others untouched
        run_variant('cv_vit_houlsby', base, images, masks, noise, make_args(), autocast_only=True)
        run_variant('cv_vit_compacter', base, images, masks, noise, make_args(adapter_type='compacter'), autocast_only=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == '--prompt-only':
        run_variant('cv_vit_prompt', base, images, masks, noise, make_args(adapter_type='prompt', n_tokens=5))
        return
    run_variant('cv_vit_houlsby', base, images, masks, noise, make_args())
    run_variant('cv_vit_houlsby_gelu_ln', base, images, masks, noise, make_args(adapter_activation='GELU'), layernorm=True)
    run_variant('cv_vit_pfeiffer_ver2', base, images, masks, noise, make_args(adapter_type='pfeiffer_ver2'))
    run_variant('cv_vit_compacter', base, images, masks, noise, make_args(adapter_type='compacter'))
    run_variant('cv_vit_cpc', base, images, masks, noise, make_args(arch='cpc'))
    run_variant('cv_mae_houlsby', base_mae, images, masks, noise, make_args(CV_model_load='vit-mae-base'))
    run_variant('cv_vit_frozen', base, images, masks, noise, make_args(adapter_type='none'))


if __name__ == '__main__':
    main()
