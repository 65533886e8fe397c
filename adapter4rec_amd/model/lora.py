"""LoRALinear: parameter container with the state_dict layout and semantics of ``loralib==0.1.1``'s ``lora.Linear``
(third party; the reference pins it in README.md:65 and calls it at Downstream/Text/run.py:414-428):

    y = x W^T + b + (x A^T B^T) * (lora_alpha / r),   lora_alpha = 1

``weight`` is frozen, ``bias`` stays trainable (loralib only freezes ``weight``), ``lora_A`` [r, in] is
kaiming-uniform(a = sqrt(5)), ``lora_B`` [out, r] zeros.  loralib is not in the image and not vendored.  Parity is pinned through the
reference's own numbers: with ``W = W_base - B A / r`` the oracle and the HIP path must reproduce the imported
reference's forward, and ``dA = B^T dW / r``, ``dB = dW A^T / r`` from its ``dW`` (tools/gen_golden_r4.py -> tests/golden/lora_pin_*.npz,
tests/test_oracle_golden.py, tests/test_engine_gpu.py).  What stays unpinnable here: loralib's own init and its ``lora_alpha`` default."""
import math

import torch
from torch import nn

from .bert import _Container


class LoRALinear(_Container):
    def __init__(self, in_features, out_features, r=0, lora_alpha=1, bias=True):
        super().__init__()
        self.in_features, self.out_features, self.r = in_features, out_features, r
        self.weight = nn.Parameter(torch.empty(out_features, in_features), requires_grad=(r == 0))   # loralib freezes it only when r > 0
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(in_features)
            nn.init.uniform_(self.bias, -bound, bound)
        if r > 0:
            self.lora_A = nn.Parameter(torch.empty(r, in_features))
            self.lora_B = nn.Parameter(torch.zeros(out_features, r))
            nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
            self.scaling = lora_alpha / r
        else:
            self.scaling = 0.0
