"""The TORCH_LIBRARY form of the boundary: ``torch.ops.a4r.*`` -- at::Tensor arguments, TORCH_CHECK errors, kernels
enqueued on the current HIP stream -- registered by ``liba4r_torch_ops.so`` (adapter4rec_amd/csrc/a4r_torch_ops.cpp), a host-only shim over
the C ABI of ``liba4r_hip.so``.  The training path itself binds the C ABI through ctypes (``_lib.py``); this module is for C++ /
TorchScript / dispatcher-level callers and for the tests that hold the two bindings to each other.

    from adapter4rec_amd import torch_ops
    ops = torch_ops.load()                                   # raises when the library has not been built
    ops.gemm_nt(x, w, y, bias, None, None, 0, 1.0, 0.0, 0, 0, False)
"""
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
OPS_LIB_PATH = os.path.join(_HERE, 'liba4r_torch_ops.so')
OPS = ('gemm_nt', 'adapter_residual_ln_fwd', 'adapter_residual_ln_bwd', 'ln_fwd', 'score_bce_fwd', 'score_bce_bwd', 'fused_adam_step',
       'topk_rank_eval', 'lora_bwd', 'encoder_layer_fwd', 'encoder_layer_bwd', 'sasrec_block_fwd', 'sasrec_block_bwd', 'embed_ln_fwd', 'patch_embed_fwd',
       'vit_assemble', 'abi_version')
_loaded = False


def load():
    global _loaded
    if not _loaded:
        if not os.path.exists(OPS_LIB_PATH):
            raise RuntimeError(f'{OPS_LIB_PATH} not found: build it with `make -C adapter4rec_amd/csrc` (python -c "import __graft_entry__ as g; g.build()")')
        torch.ops.load_library(OPS_LIB_PATH)
        from . import _lib
        got = int(torch.ops.a4r.abi_version())
        if got != _lib.ABI_VERSION:
            raise RuntimeError(f'{OPS_LIB_PATH} was built against ABI {got}, this package is {_lib.ABI_VERSION}: rebuild')
        _loaded = True
    return torch.ops.a4r
