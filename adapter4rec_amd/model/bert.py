"""Parameter container with the attribute tree / state_dict keys of HuggingFace ``BertModel`` /
``RobertaModel`` (transformers 4.20.1 layout: ``encoder.layer[i].attention.self.query`` ...), so the
reference's module-replacement injection (Downstream/Text/run.py:385-479) works on it unchanged and
checkpoints stay key-compatible.  It owns no maths: the native engine reads its tensors.
A real ``transformers.BertModel`` / ``RobertaModel`` can be passed to ``Model`` instead -- same tree.
"""
import torch
from torch import nn


def _no_module_forward(self, *a, **k):
    raise NotImplementedError(
        f'{type(self).__name__} is a parameter container: the MI355X-native engine runs the fused path. '
        'Call Model(...)(sample_items, log_mask, local_rank), model.bert_encoder(ids) or model.user_encoder(...).')


class _Container(nn.Module):
    forward = _no_module_forward


class BertSelfAttentionParams(_Container):
    def __init__(self, cfg):
        super().__init__()
        h = cfg['hidden_size']
        self.query, self.key, self.value = nn.Linear(h, h), nn.Linear(h, h), nn.Linear(h, h)
        self.dropout = nn.Dropout(cfg['attention_probs_dropout_prob'])


class BertSelfOutputParams(_Container):
    """HF BertSelfOutput / BertOutput: dense, LayerNorm, dropout."""

    def __init__(self, cfg, in_features):
        super().__init__()
        h = cfg['hidden_size']
        self.dense = nn.Linear(in_features, h)
        self.LayerNorm = nn.LayerNorm(h, eps=cfg['layer_norm_eps'])
        self.dropout = nn.Dropout(cfg['hidden_dropout_prob'])


class BertAttentionParams(_Container):
    def __init__(self, cfg):
        super().__init__()
        self.self = BertSelfAttentionParams(cfg)
        self.output = BertSelfOutputParams(cfg, cfg['hidden_size'])


class BertIntermediateParams(_Container):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg['hidden_size'], cfg['intermediate_size'])


class BertLayerParams(_Container):
    def __init__(self, cfg):
        super().__init__()
        self.attention = BertAttentionParams(cfg)
        self.intermediate = BertIntermediateParams(cfg)
        self.output = BertSelfOutputParams(cfg, cfg['intermediate_size'])


class BertEncoderParams(_Container):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([BertLayerParams(cfg) for _ in range(cfg['num_hidden_layers'])])


class BertEmbeddingsParams(_Container):
    def __init__(self, cfg):
        super().__init__()
        h = cfg['hidden_size']
        self.word_embeddings = nn.Embedding(cfg['vocab_size'], h, padding_idx=cfg.get('pad_token_id', 0))
        self.position_embeddings = nn.Embedding(cfg['max_position_embeddings'], h)
        self.token_type_embeddings = nn.Embedding(cfg['type_vocab_size'], h)
        self.LayerNorm = nn.LayerNorm(h, eps=cfg['layer_norm_eps'])
        self.dropout = nn.Dropout(cfg['hidden_dropout_prob'])


class BertPoolerParams(_Container):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg['hidden_size'], cfg['hidden_size'])


BERT_BASE = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                 max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1, pad_token_id=0, model_type='bert')
ROBERTA_BASE = dict(BERT_BASE, vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5,
                    pad_token_id=1, model_type='roberta')


class BertBackbone(_Container):
    """Random-init (HF initializer_range 0.02) BERT/RoBERTa-shaped backbone; ``config`` is a plain dict."""

    def __init__(self, config=None, **overrides):
        super().__init__()
        cfg = dict(BERT_BASE if config is None else config)
        cfg.update(overrides)
        self.config = cfg
        self.embeddings = BertEmbeddingsParams(cfg)
        self.encoder = BertEncoderParams(cfg)
        self.pooler = BertPoolerParams(cfg)
        std = cfg.get('initializer_range', 0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=std)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, std=std)
                if m.padding_idx is not None:
                    with torch.no_grad():
                        m.weight[m.padding_idx].zero_()

    def get_input_embeddings(self):              # HF API used by the soft-prompt injection (run.py:429-434)
        return self.embeddings.word_embeddings

    def set_input_embeddings(self, value):
        self.embeddings.word_embeddings = value

    @classmethod
    def from_config_json(cls, path):
        import json
        with open(path) as f:
            raw = json.load(f)
        keys = list(BERT_BASE.keys())
        return cls({k: raw[k] for k in keys if k in raw})


def backbone_geometry(bert_model):
    """(cfg dict) from either a BertBackbone or a HuggingFace model."""
    c = bert_model.config
    get = (lambda k, d=None: c.get(k, d)) if isinstance(c, dict) else (lambda k, d=None: getattr(c, k, d))
    return dict(hidden_size=get('hidden_size'), num_hidden_layers=get('num_hidden_layers'),
                num_attention_heads=get('num_attention_heads'), intermediate_size=get('intermediate_size'),
                layer_norm_eps=get('layer_norm_eps'), hidden_dropout_prob=get('hidden_dropout_prob', 0.1),
                attention_probs_dropout_prob=get('attention_probs_dropout_prob', 0.1),
                pad_token_id=get('pad_token_id', 0) or 0, model_type=get('model_type', 'bert'),
                hidden_act=get('hidden_act', 'gelu'))
