"""Seeded model + batch builders at the geometries bench.py times (pure torch on CPU: no GPU, no reference import), shared by
  * tests/test_parity_base_gpu.py   (HIP fp32 / bf16 / fp8 vs the CPU oracle on the GPU box), and
  * tools/gen_golden_r3.py          (the IMPORTED reference, fp32 and under autocast(bfloat16), on the same weights in the build
                                     container -> tests/golden/base_geom_*.npz).
The weights come from torch's CPU generator under a fixed seed, so both sides build bit-identical tensors (same torch build in
the build container and on the GPU box; the fixtures carry a checksum that the tests compare before using them)."""
import argparse

import torch

GEOMETRY = {
    'bert': dict(vocab_size=30522, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, pad_token_id=0),
    'roberta': dict(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1),
}


def text_args(dtype, act='RELU', adapter_type='houslby', arch='sasrec'):
    return argparse.Namespace(
        max_seq_len=20, l2_weight=0, embedding_dim=64, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
        num_words_title=30, num_words_abstract=50, num_words_body=50, news_attributes=['title'], word_embedding_dim=768,
        bert_model_load='bert_base_uncased', bert_adapter_down_size=64, adapter_down_size=16, adapter_dropout_rate=0.1,
        adapter_activation=act, hypercomplex_division=4, phm_init_range=1e-4, adapter_type=adapter_type, is_serial='True',
        adding_adapter_to='all', arch=arch, compute_dtype=dtype)


def build_text_case(encoder='bert', act='RELU', adapter_type='houslby', arch='sasrec', seed=3, users=2, n_items=4096, full_histories=False):
    """BERT-base / RoBERTa-base geometry (12 x 768, 12 heads, F = 3072, S = 30), `users` users = 42 item slots each: one full
    history and short ones (left-padded with the PAD item), full and partially padded titles."""
    from adapter4rec_amd.inject import freeze_all, inject_adapters
    from adapter4rec_amd.model import BERT_BASE, ROBERTA_BASE, BertBackbone, Model, ModelCPC
    torch.manual_seed(seed)
    roberta = encoder == 'roberta'
    cls = ModelCPC if arch == 'cpc' else Model
    model = cls(text_args('fp32', act, adapter_type, arch), n_items, True, BertBackbone(ROBERTA_BASE if roberta else BERT_BASE))
    freeze_all(model)
    model = inject_adapters(model, model.args)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.requires_grad:                       # adapter biases / fc_up start at 0 / 1e-2: give every gradient path a signal
                p.add_(0.02 * torch.randn_like(p))
    model.eval()
    g = torch.Generator().manual_seed(seed)
    L = 21
    bos, eos, pad = (0, 2, 1) if roberta else (101, 102, 0)
    ids = torch.zeros(users, L, 2, 60, dtype=torch.int64)
    mask = torch.zeros(users, L - 1)
    for u in range(users):
        n = L if (u == 0 or full_histories) else 9    # one full history, one short (left-padded with the PAD item); full_histories: every user full
        for slot in range(L - n, L):
            for side in range(2):
                if side == 1 and slot == L - 1:
                    continue
                ln = 30 if (slot + side) % 3 else int(torch.randint(4, 30, (1,), generator=g))      # full and partially padded titles
                if roberta:
                    ids[u, slot, side, :30] = pad      # RoBERTa pads titles with id 1; the PAD ITEM stays all zeros (preprocess: item 0)
                ids[u, slot, side, 0] = bos
                ids[u, slot, side, 1:ln - 1] = torch.randint(1000, 30000, (ln - 2,), generator=g)
                ids[u, slot, side, ln - 1] = eos
                ids[u, slot, side, 30:30 + ln] = 1
        mask[u, L - n:] = 1
    return model, ids.view(-1, 60), mask


REAL_MINI = dict(hidden_size=256, num_hidden_layers=4, num_attention_heads=4, intermediate_size=1024)


def build_real_case(item_num, seed=5):
    """The model of tests/golden/real_batch.npz (tools/gen_golden_r6.py): BERT-mini geometry (4 x 256, 4 heads, F = 1024: run.py:100-114's
    `bert_mini`) with the REAL 30 522-entry vocabulary, Houlsby adapters (GELU: smooth, so gradients hold 1e-4), seeded weights.  The batch is
    not built here: it is the fixture's (real Adressa titles, real Amazon histories)."""
    from adapter4rec_amd.inject import freeze_all, inject_adapters
    from adapter4rec_amd.model import BERT_BASE, BertBackbone, Model
    torch.manual_seed(seed)
    args = text_args('fp32', 'GELU')
    args.word_embedding_dim, args.bert_model_load = 256, 'bert_mini_uncased'
    model = Model(args, item_num, True, BertBackbone(dict(BERT_BASE, **REAL_MINI)))
    freeze_all(model)
    model = inject_adapters(model, model.args)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.requires_grad:
                p.add_(0.02 * torch.randn_like(p))
    model.eval()
    return model


def checksum(model):
    """Order-dependent fp64 checksum of every tensor of the state dict (fixture <-> rebuilt weights)."""
    tot = 0.0
    for i, (k, v) in enumerate(model.state_dict().items()):
        tot += (i % 7 + 1) * float(v.double().sum())
    return tot


def cv_args(dtype, kind, lora_r=8):
    mae = kind == 'mae_compacter'
    return argparse.Namespace(
        max_seq_len=20, l2_weight=0, embedding_dim=64, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
        CV_model_load='vit-mae-base' if mae else 'vit-base-patch16-224', CV_resize=224,
        cv_adapter_down_size=64, adapter_down_size=16, adapter_dropout_rate=0.1, adapter_activation='RELU',
        hypercomplex_division=4, phm_init_range=1e-4, adapter_type='compacter' if mae else 'lora',
        is_serial='True', adding_adapter_to='all', arch='sasrec', compute_dtype=dtype, lora_r=lora_r, lora_r_sasrec=4)


def build_vit_case(kind='vit_lora', seed=7, users=1, lora_r=8):
    """configs[2] / configs[4] at the benchmarked geometry: ViT-B/16 (768 x 12, 224 x 224 images -> 197 tokens) + LoRA r = 8, or
    ViT-MAE-base (75 % masked -> 50 tokens) + Compacter; `users` users = 42 image slots each (uint8 HWC pixels)."""
    from adapter4rec_amd.cv import Model, ViTForImageClassification, ViTMAEModel
    from adapter4rec_amd.cv.inject import inject_adapters
    from adapter4rec_amd.inject import freeze_all
    torch.manual_seed(seed)
    args = cv_args('fp32', kind, lora_r)
    if kind == 'mae_compacter':
        net = ViTMAEModel()
    else:
        net = ViTForImageClassification(num_labels=args.embedding_dim)
        torch.nn.init.xavier_normal_(net.classifier.weight)
    model = Model(args, 512, True, net)
    freeze_all(model)
    root = inject_adapters(model, args)
    with torch.no_grad():
        for n, p in root.named_parameters():
            if p.requires_grad:                       # lora_B starts at 0, phm tensors at 1e-4: every gradient path gets a signal
                p.add_((0.05 if kind == 'mae_compacter' else 0.02) * torch.randn_like(p))
            if n.endswith('image_net.classifier.weight') or n.endswith('cv_proj.weight'):
                p.mul_(0.25)                          # xavier-init item head on unit-variance LayerNorm rows gives |score| ~ 25 (a saturated
                                                      # sigmoid: any rounding is amplified exponentially); O(1) scores = the regime of a trained model
    root.eval()
    g = torch.Generator().manual_seed(seed)
    n = users * 42
    img = torch.randint(0, 256, (users, 21, 2, 224, 224, 3), generator=g, dtype=torch.uint8)
    mask = torch.ones(users, 20)
    if users > 1:
        mask[1, :11] = 0
    img[:, -1, 1] = 0
    noise = torch.rand(n, 196, generator=g) if kind == 'mae_compacter' else None
    return root, img.view(n, 224, 224, 3), mask, noise
