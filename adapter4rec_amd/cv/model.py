"""Model / ModelCPC, Vit_Encoder / MAE_Encoder and the ViT adapter wrappers with the constructor and call signatures of
the reference's image path (Downstream/CV/model/model.py, encoders.py) -- the drop-in boundary for configs 3 and 5.

    cv_model = ViTForImageClassification(...); cv_model.classifier = nn.Linear(768, args.embedding_dim)   # run_adapter.py:289-296
    model = Model(args, item_num, use_modal, cv_model)                                                    # :338
    layer.attention.output = VITAdaptedSelfOutput(layer.attention.output, args)                          # :428-434
    loss = model(sample_items.view(-1, 3, R, R), log_mask, local_rank); loss.backward()                   # :582-594

``sample_items`` may also be raw uint8 pixels [n, R, R, 3] (the LMDB record content, data_utils/dataset.py:17-27): the
ToTensor + Normalize(0.5, 0.5) half of the reference's CPU transform then runs inside the patch kernel."""
import torch
from torch import nn
from torch.nn.init import constant_, xavier_normal_

from ..model.bert import _Container
from ..model.encoders import User_Encoder
from ..model.model import (_NativeLoss, CompacterModel, SASRecAdaptedSelfOutput, SASRecCompacterAdaptedSelfOutput,   # noqa: F401
                           SASRecParallelAdaptedSelfOutput, SASRecPfeifferVer2AdaptedSelfOutput)
from ..model.model import SASRecKAdaptedTransformerBlocks  # noqa: F401
from ..model.modules import AdapterBlock, HyperComplexAdapterBlock, KAdapterBlock


class Vit_Encoder(nn.Module):                    # encoders.py:25-32
    def __init__(self, image_net):
        super().__init__()
        self.image_net = image_net
        self.activate = nn.GELU()
        self._owner = [None]

    def forward(self, item_content):
        """[n, 3, R, R] fp32 (or [n, R, R, 3] uint8) -> [n, E] item embeddings (inference; eval uses it, metrics.py)."""
        return self._owner[0]._engine().encode_items(item_content)


class MAE_Encoder(nn.Module):                    # encoders.py:8-22
    def __init__(self, image_net, item_dim):
        super().__init__()
        self.item_dim, self.word_emb = item_dim, 768
        self.image_net = image_net
        self.activate = nn.GELU()
        self.cv_proj = nn.Linear(image_net.config['hidden_size'] if isinstance(image_net.config, dict) else image_net.config.hidden_size,
                                 item_dim)
        xavier_normal_(self.cv_proj.weight.data)
        constant_(self.cv_proj.bias.data, 0)
        self._owner = [None]

    def forward(self, item_content, noise=None):
        return self._owner[0]._engine().encode_items(item_content, noise=noise)


class _CVBase(nn.Module):
    arch = 'sasrec'

    def __init__(self, args, item_num, use_modal, image_net):
        super().__init__()
        if not use_modal:
            raise NotImplementedError('ID tower (use_modal=False) is out of scope')
        self.args = args
        self.use_modal = use_modal
        self.max_seq_len = args.max_seq_len
        self.l2_weight = args.l2_weight / 2
        self.user_encoder = User_Encoder(item_num=item_num, max_seq_len=args.max_seq_len, item_dim=args.embedding_dim,
                                         num_attention_heads=args.num_attention_heads, dropout=args.drop_rate,
                                         n_layers=args.transformer_block)
        name = args.CV_model_load
        if 'mae' in name:                        # model.py:29-30
            self.cv_encoder = MAE_Encoder(image_net=image_net, item_dim=args.embedding_dim)
        elif 'vit' in name:
            self.cv_encoder = Vit_Encoder(image_net=image_net)
        else:
            raise NotImplementedError(f'--CV_model_load {name}: the native image tower covers ViT and ViT-MAE (resnet/beit/swin are not in the BASELINE configs)')
        self.criterion = nn.BCEWithLogitsLoss()
        self.cv_encoder._owner[0] = self
        self.user_encoder._owner[0] = self
        self._native = [None]
        self._phm_owner = [None]
        self.compute_dtype = getattr(args, 'compute_dtype', 'bf16')
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_native())

    def invalidate_native(self):
        self._native[0] = None

    def _engine(self):
        if self._native[0] is None:
            from ..engine_vit import ViTRecEngine
            self._native[0] = ViTRecEngine(self, self.args, arch=self.arch, dtype=self.compute_dtype, phm_owner=self._phm_owner[0])
        return self._native[0]

    def item_encoder_in(self, dtype):
        """items -> embeddings computed in `dtype` on the current weights (a forward-only snapshot engine; eval's item sweep in
        fp32 while training runs in bf16: data_utils/metrics.py get_item_embeddings, --eval_compute_dtype)."""
        if dtype == self.compute_dtype:
            return self._engine().encode_items
        from ..engine_vit import ViTRecEngine
        snap = ViTRecEngine.inference_snapshot(self, self.args, self.arch, dtype, self._phm_owner[0])
        return snap.encode_items

    def _apply(self, fn, *a, **k):
        self.invalidate_native()
        return super()._apply(fn, *a, **k)

    def forward(self, sample_items, log_mask, local_rank=None, noise=None):
        eng = self._engine()
        eng.next_noise = noise                   # ViT-MAE: explicit masking noise [n, n_patches] (parity runs); None = drawn on device
        if not log_mask.is_cuda:                 # host log_mask: the batch's pad structure, read without a sync (model/model.py)
            eng.host_log_mask = log_mask
            log_mask = log_mask.to(eng.dev, non_blocking=True)
        else:
            eng.host_log_mask = None
        if torch.is_grad_enabled() and eng.n_trainable:
            return _NativeLoss.apply(eng, sample_items, log_mask, *eng.trainable_params)
        return eng.train_forward(sample_items, log_mask)


class Model(_CVBase):                            # model.py:10-77
    arch = 'sasrec'


class ModelCPC(_CVBase):                         # model.py:80-146
    arch = 'cpc'


class SoftPrompt(_Container):                    # model.py:512-535: n_tokens learned rows appended AFTER cls + patches (no position term)
    def __init__(self, wte, n_tokens=10, embed_dim=768):
        super().__init__()
        self.wte = wte
        self.patch_embeddings = wte.patch_embeddings
        self.n_tokens = n_tokens
        self.Prompt_Tokens = nn.Parameter(torch.zeros(1, n_tokens, embed_dim))


class VITKAdaptedCVModel(_Container):            # model.py:374-404 (K-Adapter on the image tower)
    """Sits where ``image_net.vit.encoder`` was (run_adapter.py:378-380): KAdapterBlock j reads hidden_states[i + 1] for the j-th i of
    --k_adapter_bert_list plus the previous adapter's output; ``com_dense`` fuses [last hidden state ; last adapter output] back to
    the ViT width before ``vit.layernorm``.  Keys: ``...vit.encoder.vit_encoder.layer.*``, ``...bert_adapter_list.*``, ``...com_dense.*``."""

    def __init__(self, vit_encoder, args):
        super().__init__()
        dim = vit_encoder.layer[0].layernorm_before.normalized_shape[0]       # the reference hard-codes 768 (ViT-B)
        self.vit_encoder = vit_encoder
        self.k_adapter_num_list = [int(i) + 1 for i in str(args.k_adapter_bert_list).split(',')]
        self.bert_adapter_list = nn.ModuleList([KAdapterBlock(args, args.num_adapter_heads_bert, dim, args.k_adapter_bert_hidden_dim,
                                                              args.adapter_dropout_rate) for _ in self.k_adapter_num_list])
        self.com_dense = nn.Linear(dim * 2, dim)

    @property
    def layer(self):
        return self.vit_encoder.layer


# ---------------------------------------------------------------- ViT-side wrappers (pre-LN: no LayerNorm inside)
class VITAdaptedSelfOutput(_Container):          # model.py:182-195: adapter(dropout(dense(x)))          (residual added by ViTLayer)
    placement, adds_input = 'serial', False

    def __init__(self, self_output, args):
        super().__init__()
        self.self_output = self_output
        # the reference hard-codes 768 (ViT-B); the dense layer's own width is the same number there and also serves small test geometries
        self.adapter = AdapterBlock(args, self_output.dense.out_features, args.cv_adapter_down_size, args.adapter_dropout_rate)


class VITAdaptedOutput(VITAdaptedSelfOutput):    # model.py:198-212: adapter(dropout(dense(x))) + input
    adds_input = True


class VITAdaptedParallelOutput(VITAdaptedSelfOutput):   # model.py:165-179: dense(x) + input + adapter(input) (adapter keeps its inner residual)
    placement, adds_input = 'parallel', True


class VITCompacterAdaptedSelfOutput(_Container):   # model.py:432-445
    placement, adds_input = 'serial', False

    def __init__(self, self_output, args):
        super().__init__()
        self.self_output = self_output
        self.adapter = HyperComplexAdapterBlock(args, self_output.dense.out_features, args.cv_adapter_down_size)


class VITCompacterAdaptedOutput(VITCompacterAdaptedSelfOutput):   # model.py:448-462
    adds_input = True
