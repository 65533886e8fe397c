"""Parameter containers with the attribute tree / state_dict keys of HuggingFace ``ViTForImageClassification`` and
``ViTMAEModel`` in the transformers==4.20.1 layout the reference addresses (Downstream/CV/run_adapter.py:286-297,
369-372, 384-395, 428-434): ``vit.encoder.layer[i].attention.attention.{query,key,value}``, ``.attention.output``
(dense + dropout, NO residual / LayerNorm: ViT is pre-LN), ``.intermediate.dense``, ``.output`` (dense + dropout;
the residual lives in ViTLayer), ``.layernorm_before/after``, ``vit.layernorm``, ``classifier``.
No maths lives here: the native engine (adapter4rec_amd/engine_vit.py) reads the tensors."""
import math

import torch
from torch import nn

from ..model.bert import _Container

VIT_BASE = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, image_size=224,
                patch_size=16, num_channels=3, layer_norm_eps=1e-12, hidden_dropout_prob=0.0,
                attention_probs_dropout_prob=0.0, hidden_act='gelu', initializer_range=0.02, num_labels=1000,
                model_type='vit', mask_ratio=0.75)


class ViTSelfAttentionParams(_Container):
    def __init__(self, cfg):
        super().__init__()
        h = cfg['hidden_size']
        self.query, self.key, self.value = nn.Linear(h, h), nn.Linear(h, h), nn.Linear(h, h)
        self.dropout = nn.Dropout(cfg['attention_probs_dropout_prob'])


class ViTSelfOutputParams(_Container):          # ViTSelfOutput / ViTOutput: dense + dropout
    def __init__(self, cfg, in_features):
        super().__init__()
        self.dense = nn.Linear(in_features, cfg['hidden_size'])
        self.dropout = nn.Dropout(cfg['hidden_dropout_prob'])


class ViTAttentionParams(_Container):
    def __init__(self, cfg):
        super().__init__()
        self.attention = ViTSelfAttentionParams(cfg)
        self.output = ViTSelfOutputParams(cfg, cfg['hidden_size'])


class ViTIntermediateParams(_Container):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg['hidden_size'], cfg['intermediate_size'])


class ViTLayerParams(_Container):
    def __init__(self, cfg):
        super().__init__()
        h, eps = cfg['hidden_size'], cfg['layer_norm_eps']
        self.attention = ViTAttentionParams(cfg)
        self.intermediate = ViTIntermediateParams(cfg)
        self.output = ViTSelfOutputParams(cfg, cfg['intermediate_size'])
        self.layernorm_before = nn.LayerNorm(h, eps=eps)
        self.layernorm_after = nn.LayerNorm(h, eps=eps)


class ViTEncoderParams(_Container):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([ViTLayerParams(cfg) for _ in range(cfg['num_hidden_layers'])])


class ViTPatchEmbeddingsParams(_Container):
    def __init__(self, cfg):
        super().__init__()
        p = cfg['patch_size']
        self.projection = nn.Conv2d(cfg['num_channels'], cfg['hidden_size'], kernel_size=p, stride=p)


class ViTEmbeddingsParams(_Container):
    def __init__(self, cfg, learned_pos=True):
        super().__init__()
        h = cfg['hidden_size']
        n = (cfg['image_size'] // cfg['patch_size']) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, h))
        self.patch_embeddings = ViTPatchEmbeddingsParams(cfg)
        self.position_embeddings = nn.Parameter(torch.zeros(1, n + 1, h), requires_grad=learned_pos)
        self.dropout = nn.Dropout(cfg['hidden_dropout_prob'])


def _init(module, std):
    for m in module.modules():
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            nn.init.trunc_normal_(m.weight, std=std)
            nn.init.zeros_(m.bias)


class ViTModelParams(_Container):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = ViTEmbeddingsParams(cfg)
        self.encoder = ViTEncoderParams(cfg)
        self.layernorm = nn.LayerNorm(cfg['hidden_size'], eps=cfg['layer_norm_eps'])


class ViTForImageClassification(_Container):
    """``image_net`` of Vit_Encoder: ``.vit`` + ``.classifier`` (replaced by Linear(768, embedding_dim), run_adapter.py:291-296)."""

    def __init__(self, config=None, **overrides):
        super().__init__()
        cfg = dict(VIT_BASE)
        cfg.update(config or {})
        cfg.update(overrides)
        self.config = cfg
        self.vit = ViTModelParams(cfg)
        self.classifier = nn.Linear(cfg['hidden_size'], cfg['num_labels'])
        _init(self, cfg['initializer_range'])
        with torch.no_grad():
            nn.init.trunc_normal_(self.vit.embeddings.position_embeddings, std=cfg['initializer_range'])
            nn.init.trunc_normal_(self.vit.embeddings.cls_token, std=cfg['initializer_range'])


def sincos_2d(h, grid):
    """ViT-MAE's fixed 2-D sin-cos position table [1 + grid*grid, h] (row 0 = cls = zeros), as HF get_2d_sincos_pos_embed."""
    def one(d, pos):
        om = 1.0 / 10000 ** (torch.arange(d // 2, dtype=torch.float64) / (d / 2.0))
        out = pos.reshape(-1)[:, None] * om[None]
        return torch.cat([out.sin(), out.cos()], 1)
    gh = torch.arange(grid, dtype=torch.float64)
    gw = torch.arange(grid, dtype=torch.float64)
    w, hh = torch.meshgrid(gw, gh, indexing='xy')            # np.meshgrid(grid_w, grid_h): w first
    emb = torch.cat([one(h // 2, w), one(h // 2, hh)], 1)
    return torch.cat([torch.zeros(1, h, dtype=torch.float64), emb], 0).float()


class ViTMAEModel(_Container):
    """``image_net`` of MAE_Encoder (ViTMAEModel: embeddings + encoder + layernorm; 75 % of the patches are dropped by
    argsort of uniform noise before the encoder)."""

    def __init__(self, config=None, **overrides):
        super().__init__()
        cfg = dict(VIT_BASE, model_type='vit_mae')
        cfg.update(config or {})
        cfg.update(overrides)
        self.config = cfg
        self.embeddings = ViTEmbeddingsParams(cfg, learned_pos=False)
        self.encoder = ViTEncoderParams(cfg)
        self.layernorm = nn.LayerNorm(cfg['hidden_size'], eps=cfg['layer_norm_eps'])
        _init(self, cfg['initializer_range'])
        with torch.no_grad():
            grid = cfg['image_size'] // cfg['patch_size']
            self.embeddings.position_embeddings.copy_(sincos_2d(cfg['hidden_size'], grid)[None])
            nn.init.normal_(self.embeddings.cls_token, std=cfg['initializer_range'])
            w = self.embeddings.patch_embeddings.projection.weight
            nn.init.xavier_uniform_(w.view(w.shape[0], -1))


def vit_geometry(image_net):
    c = image_net.config
    get = (lambda k, d=None: c.get(k, d)) if isinstance(c, dict) else (lambda k, d=None: getattr(c, k, d))
    return dict(hidden_size=get('hidden_size'), num_hidden_layers=get('num_hidden_layers'),
                num_attention_heads=get('num_attention_heads'), intermediate_size=get('intermediate_size'),
                image_size=get('image_size'), patch_size=get('patch_size'), num_channels=get('num_channels', 3),
                layer_norm_eps=get('layer_norm_eps'), hidden_dropout_prob=get('hidden_dropout_prob', 0.0),
                attention_probs_dropout_prob=get('attention_probs_dropout_prob', 0.0), hidden_act=get('hidden_act', 'gelu'),
                mask_ratio=get('mask_ratio', 0.75), model_type=get('model_type', 'vit'))
