"""Parameter containers mirroring the reference's SASRec and adapter modules
(Downstream/Text/model/modules.py, layers.py): same attribute names => same state_dict keys, same
initial distributions.  No maths lives here; the native engine consumes the tensors."""
import math

import torch
from torch import nn

from .bert import _Container


class PositionwiseFeedForward(_Container):      # modules.py:16-28
    def __init__(self, d_model, d_inner, dropout):
        super().__init__()
        self.w_1 = nn.Linear(d_model, d_inner)
        self.w_2 = nn.Linear(d_inner, d_model)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.dropout = nn.Dropout(dropout)


class MultiHeadedAttention(_Container):         # modules.py:45-74
    def __init__(self, n_heads, d_model, dropout):
        super().__init__()
        assert d_model % n_heads == 0
        self.d_model, self.n_heads = d_model, n_heads
        self.d_k = self.d_v = d_model // n_heads
        self.w_Q = nn.Linear(d_model, d_model, bias=False)
        self.w_K = nn.Linear(d_model, d_model, bias=False)
        self.w_V = nn.Linear(d_model, d_model, bias=False)
        self.fc = nn.Linear(d_model, d_model, bias=False)
        self.dropout = nn.Dropout(p=dropout)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)


class TransformerBlock(_Container):             # modules.py:77-87
    def __init__(self, d_model, n_heads, d_inner, dropout):
        super().__init__()
        self.multi_head_attention = MultiHeadedAttention(n_heads=n_heads, d_model=d_model, dropout=dropout)
        self.feed_forward = PositionwiseFeedForward(d_model=d_model, d_inner=d_inner, dropout=dropout)


class TransformerEncoder(_Container):           # modules.py:90-113
    def __init__(self, n_vocab, n_position, d_model, n_heads, dropout, n_layers):
        super().__init__()
        self.position_embedding = nn.Embedding(n_position, d_model)
        self.dropout = nn.Dropout(p=dropout)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.transformer_blocks = nn.ModuleList(
            [TransformerBlock(d_model=d_model, n_heads=n_heads, d_inner=d_model * 4, dropout=dropout) for _ in range(n_layers)])


class AdapterBlock(_Container):                 # modules.py:116-134 (Houlsby: inner residual, N(0,1e-2) init)
    kind = 'houlsby'

    def __init__(self, args, input_size, down_size, dropout=0.1):
        super().__init__()
        self.fc_down = nn.Linear(input_size, down_size)
        self.fc_up = nn.Linear(down_size, input_size)
        for lin in (self.fc_down, self.fc_up):
            nn.init.normal_(lin.weight, std=1e-2)
            nn.init.zeros_(lin.bias)
        self.activation_name = 'GELU' if args.adapter_activation == 'GELU' else 'relu'
        self.dropout = nn.Dropout(dropout)      # constructed, never applied by the reference (modules.py:129-134)


class AdapterPfeifferBlock(_Container):         # modules.py:137-158 (no inner residual, default Linear init)
    kind = 'pfeiffer'

    def __init__(self, args, input_size, down_size, dropout=0.1):
        super().__init__()
        self.fc_down = nn.Linear(input_size, down_size)
        self.fc_up = nn.Linear(down_size, input_size)
        if args.adapter_activation not in ('GELU', 'leaky_relu', 'relu'):
            # the reference creates no `activate` attribute in this case and fails at the first forward
            raise AttributeError("AdapterPfeifferBlock needs --adapter_activation in {GELU, leaky_relu, relu}")
        self.activation_name = args.adapter_activation
        self.dropout = nn.Dropout(dropout)


class KAdapterBlock(_Container):                # modules.py:161-206 (K-Adapter: down -> 2 plain transformer blocks -> up, + input)
    def __init__(self, args, num_head, input_size, down_size, dropout=0.1):
        super().__init__()
        self.num_head = num_head
        self.down_project = nn.Linear(input_size, down_size)
        self.up_project = nn.Linear(down_size, input_size)
        for lin in (self.down_project, self.up_project):
            nn.init.normal_(lin.weight, mean=0.0, std=2e-4)
            nn.init.zeros_(lin.bias)
        self.transformer_blocks = nn.ModuleList(
            [TransformerBlock(d_model=down_size, n_heads=num_head, d_inner=down_size * 4, dropout=dropout) for _ in range(2)])


class PHMLinear(_Container):                    # layers.py:25-166 in the configuration modules.py:220-249 uses
    def __init__(self, in_features, out_features, phm_dim):
        super().__init__()
        assert in_features % phm_dim == 0 and out_features % phm_dim == 0
        self.in_features, self.out_features, self.phm_dim = in_features, out_features, phm_dim
        self.W_left = nn.Parameter(torch.empty(phm_dim, in_features // phm_dim, 1))
        self.W_right = nn.Parameter(torch.empty(phm_dim, 1, out_features // phm_dim))
        self.b = nn.Parameter(torch.zeros(out_features))
        for i in range(phm_dim):                # glorot-uniform with gain sqrt(2) per slice (inits.py:10-11)
            nn.init.xavier_uniform_(self.W_left.data[i], gain=math.sqrt(2))
            nn.init.xavier_uniform_(self.W_right.data[i], gain=math.sqrt(2))
        self.phm_rule = None                    # shared rule, attached by CompacterModel (run.py:70-83)

    def set_phm_rule(self, phm_rule=None, **_):
        self.phm_rule = phm_rule


class HyperComplexAdapterBlock(_Container):     # modules.py:209-252
    kind = 'compacter'

    def __init__(self, args, input_size, down_size):
        super().__init__()
        self.down_sampler = PHMLinear(input_size, down_size, args.hypercomplex_division)
        self.up_sampler = PHMLinear(down_size, input_size, args.hypercomplex_division)
        self.activation_name = 'gelu_new'
