"""Model / ModelCPC and the adapter wrapper classes with the reference's constructor and call
signatures (Downstream/Text/model/model.py) -- the drop-in boundary of SURVEY.md section 8(b).

    model = Model(args, item_num, use_modal, bert_model)             # model.py:9-31
    layer.attention.output = BertAdaptedSelfOutput(layer.attention.output, args)   # run.py:456-460
    blocks[i] = SASRecAdaptedSelfOutput(blocks[i], args)             # run.py:462-465
    loss = model(sample_items, log_mask, local_rank); loss.backward()               # run.py:597-599

Forward/backward run on the MI355X-native engine (adapter4rec_amd/engine.py); there is no eager
PyTorch path behind these classes.
"""
import numpy as np
import torch
from torch import nn
from torch.nn.init import xavier_normal_

from .bert import _Container
from .encoders import Bert_Encoder, User_Encoder
from .modules import AdapterBlock, AdapterPfeifferBlock, HyperComplexAdapterBlock, KAdapterBlock, PHMLinear


def _bert_dim(args):
    name = args.bert_model_load
    for key, dim in (('tiny', 128), ('mini', 256), ('medium', 512), ('base', 768), ('large', 1024)):
        if key in name:
            return dim
    raise AssertionError('The pretrained model name should be defined correctly. such as bert-base-uncased so on')


class _NativeLoss(torch.autograd.Function):
    """loss = engine(sample_items, log_mask); backward() fills the adapter gradients from the native backward."""

    @staticmethod
    def forward(ctx, engine, sample_items, log_mask, *params):
        ctx.engine = engine
        ctx.n = len(params)
        return engine.train_forward(sample_items, log_mask)

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.engine
        if eng._fused_opt is not None:           # FusedAdam bound: p.grad are views of the flat gradient buffer, filled in place
            eng.backward_bound(grad_out)
            return (None, None, None) + (None,) * ctx.n
        grads = eng.train_backward(grad_out)
        return (None, None, None) + tuple(grads)


class _TransRecBase(nn.Module):
    arch = 'sasrec'

    def __init__(self, args, item_num, use_modal, bert_model):
        super().__init__()
        if not use_modal:
            raise NotImplementedError('ID tower (use_modal=False) is out of scope: Downstream/Text hard-codes is_use_modal=True (run.py:687)')
        self.args = args
        self.use_modal = use_modal
        self.max_seq_len = args.max_seq_len + 1
        self.l2_weight = args.l2_weight / 2
        self.bert_encoder = Bert_Encoder(args=args, bert_model=bert_model)
        self.user_encoder = User_Encoder(item_num=item_num, max_seq_len=args.max_seq_len, item_dim=args.embedding_dim,
                                         num_attention_heads=args.num_attention_heads, dropout=args.drop_rate,
                                         n_layers=args.transformer_block)
        self.criterion = nn.BCEWithLogitsLoss()
        self.bert_encoder._owner[0] = self
        self.user_encoder._owner[0] = self
        self._native = [None]          # engine, built lazily after adapter injection / .to(device)
        self._phm_owner = [None]       # CompacterModel holding the shared phm_rule
        self.compute_dtype = getattr(args, 'compute_dtype', 'bf16')
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_native())

    # -- engine plumbing
    def invalidate_native(self):
        self._native[0] = None

    def _engine(self):
        if self._native[0] is None:
            from ..engine import TransRecEngine
            self._native[0] = TransRecEngine(self, self.args, arch=self.arch, dtype=self.compute_dtype,
                                             phm_owner=self._phm_owner[0])
        return self._native[0]

    def item_encoder_in(self, dtype):
        """items -> embeddings computed in `dtype` on the current weights (a forward-only snapshot engine; eval's item sweep in
        fp32 while training runs in bf16: data_utils/metrics.py get_item_embeddings, --eval_compute_dtype)."""
        if dtype == self.compute_dtype:
            return self._engine().encode_items
        from ..engine import TransRecEngine
        snap = TransRecEngine.inference_snapshot(self, self.args, self.arch, dtype, self._phm_owner[0])
        return snap.encode_items

    def _apply(self, fn, *a, **k):      # .to(device) / .cuda() move tensors => rebuild the packed copies
        self.invalidate_native()
        return super()._apply(fn, *a, **k)

    def forward(self, sample_items, log_mask, local_rank=None):
        eng = self._engine()
        eng.host_max_tokens = None
        if not sample_items.is_cuda and sample_items.dim() == 2 and sample_items.dtype == torch.int64 and getattr(eng, 'n_attr', 1) > 1:
            eng.host_lens = None                   # several news attributes per row: no single title length to read
            sample_items = sample_items.to(eng.dev, non_blocking=True)
        elif not sample_items.is_cuda and sample_items.dim() == 2 and sample_items.dtype == torch.int64:
            # the batch still on the host (run.py): the longest title among its items is read here -- a training step then runs on that many tokens
            # per item instead of --num_words_title (pad tokens never reach the CLS output) -- and the rows are uploaded
            # (numpy, not torch: a torch CPU reduction wakes the whole intra-op thread pool -- 128 threads on the GPU boxes -- and cost ~19 ms per step)
            # The bound is the LAST attended position (not the count of mask ones): a left-padded tokenizer, a mask with holes or a non-0/1 mask
            # keeps every attended token, whatever its column.
            eng.host_lens = None
            if sample_items.shape[0]:
                m = sample_items.numpy()[:, sample_items.shape[1] // 2:] != 0
                last = (m * np.arange(1, m.shape[1] + 1, dtype=np.int64)).max(1)
                eng.host_max_tokens = int(last.max())
                if (m.sum(1) == last).all():       # every mask a contiguous prefix: the titles can be packed (engine.py: train_forward); the pad item counts one token
                    eng.host_lens = np.maximum(last, 1).astype(np.int32)
            sample_items = sample_items.to(eng.dev, non_blocking=True)
        if not log_mask.is_cuda:
            # log_mask still on the host (run.py hands over the DataLoader's tensor): the engine reads the batch's pad structure from it WITHOUT a
            # device synchronisation -- the item slots of short histories that the loss never reads are not encoded (engine.py train_forward)
            eng.host_log_mask = log_mask
            log_mask = log_mask.to(sample_items.device, non_blocking=True)
        else:
            eng.host_log_mask = None
        if torch.is_grad_enabled() and eng.n_trainable:
            return _NativeLoss.apply(eng, sample_items, log_mask, *eng.trainable_params)
        return eng.train_forward(sample_items, log_mask)


class Model(_TransRecBase):             # model.py:9-70
    arch = 'sasrec'


class ModelCPC(_TransRecBase):          # model.py:73-135
    arch = 'cpc'


class CompacterModel(nn.Module):        # Downstream/Text/run.py:70-83
    def __init__(self, args, model):
        super().__init__()
        n = args.hypercomplex_division
        self.model = model
        self.phm_rule = nn.Parameter(torch.empty(n, n, n).normal_(mean=0, std=args.phm_init_range))
        for _, sub in model.named_modules():
            if isinstance(sub, PHMLinear):
                sub.set_phm_rule(phm_rule=self.phm_rule)
        model._phm_owner[0] = self
        model.invalidate_native()

    def forward(self, sample_items, log_mask, local_rank=None):
        return self.model(sample_items, log_mask, local_rank)


class BertKAdaptedBertModel(_Container):        # model.py:523-559
    """Wraps the backbone: K-Adapter blocks read hidden_states[i + 1] for i in --k_adapter_bert_list (chained: each adds the previous
    adapter's output to its input), com_dense fuses [last_hidden ; last adapter output] back to the hidden width."""

    def __init__(self, bert_model, args):
        super().__init__()
        dim = _bert_dim(args)
        self.bert_model = bert_model
        self.k_adapter_num_list = [int(i) + 1 for i in str(args.k_adapter_bert_list).split(',')]
        self.bert_adapter_list = nn.ModuleList([KAdapterBlock(args, args.num_adapter_heads_bert, dim, args.k_adapter_bert_hidden_dim,
                                                              args.adapter_dropout_rate) for _ in self.k_adapter_num_list])
        self.com_dense = nn.Linear(dim * 2, dim)

    @property
    def config(self):
        return self.bert_model.config


class SASRecKAdaptedTransformerBlocks(_Container):   # model.py:562-583
    """Replaces the ModuleList of SASRec blocks: adapter i reads (block input i + previous adapter output); com_dense2 fuses
    [last block output ; last adapter output]."""

    def __init__(self, transformer_blocks, args):
        super().__init__()
        self.transformer_blocks = transformer_blocks
        self.len_transformer_blocks = len(transformer_blocks)
        self.adapter_list = nn.ModuleList([KAdapterBlock(args, args.num_adapter_heads_sasrec, args.embedding_dim, args.adapter_down_size,
                                                         args.drop_rate) for _ in range(self.len_transformer_blocks)])
        self.com_dense2 = nn.Linear(args.embedding_dim * 2, args.embedding_dim)


class SoftEmbedding(_Container):                # model.py:586-630 (soft prompt)
    """Replaces the backbone's word embedding: the first ``n_tokens`` word vectors of every title are the rows of
    ``learned_embedding`` (initialised from the first vocabulary rows), the rest are ``wte`` look-ups of tokens[:, n_tokens:]."""

    def __init__(self, wte, n_tokens=100, random_range=0.5, initialize_from_vocab=True):
        super().__init__()
        self.wte, self.n_tokens = wte, n_tokens
        if initialize_from_vocab:
            init = wte.weight[:n_tokens].clone().detach()
        else:
            init = torch.empty(n_tokens, wte.weight.size(1)).uniform_(-random_range, random_range)
        self.learned_embedding = nn.Parameter(init)


# ---------------------------------------------------------------- BERT-side wrappers
class BertAdaptedSelfOutput(_Container):            # model.py:273-297 (Houlsby, serial)
    placement = 'serial'

    def __init__(self, self_output, args):
        super().__init__()
        self.self_output = self_output
        self.adapter = AdapterBlock(args, _bert_dim(args), args.bert_adapter_down_size, args.adapter_dropout_rate)


class BertAdaptedParallelSelfOutput(BertAdaptedSelfOutput):   # model.py:246-270
    placement = 'parallel'


class BertPfeifferAdaptedSelfOutput(_Container):    # model.py:300-329
    placement = 'pfeiffer'

    def __init__(self, self_output, args):
        super().__init__()
        self.self_output = self_output
        self.adapter = AdapterPfeifferBlock(args, _bert_dim(args), args.bert_adapter_down_size, args.adapter_dropout_rate)
        self.LN = nn.LayerNorm(_bert_dim(args), eps=1e-06)


class BertCompacterAdaptedSelfOutput(_Container):   # model.py:696-720
    placement = 'serial'

    def __init__(self, self_output, args):
        super().__init__()
        self.self_output = self_output
        self.adapter = HyperComplexAdapterBlock(args, _bert_dim(args), args.bert_adapter_down_size)


# ---------------------------------------------------------------- SASRec-side wrappers
class SASRecAdaptedSelfOutput(_Container):          # model.py:332-376
    placement = 'serial'

    def __init__(self, transformer_block, args):
        super().__init__()
        self.transformer_block = transformer_block
        self.adapter1 = AdapterBlock(args, args.embedding_dim, args.adapter_down_size, args.adapter_dropout_rate)
        self.adapter2 = AdapterBlock(args, args.embedding_dim, args.adapter_down_size, args.adapter_dropout_rate)


class SASRecParallelAdaptedSelfOutput(SASRecAdaptedSelfOutput):   # model.py:474-520
    placement = 'parallel'


class SASRecPfeifferVer2AdaptedSelfOutput(_Container):   # model.py:379-423 (adapter after attention only)
    placement = 'serial'

    def __init__(self, transformer_block, args):
        super().__init__()
        self.transformer_block = transformer_block
        self.adapter1 = AdapterBlock(args, args.embedding_dim, args.adapter_down_size, args.adapter_dropout_rate)


class SASRecPfeifferAdaptedSelfOutput(_Container):  # model.py:426-471
    placement = 'pfeiffer'

    def __init__(self, transformer_block, args):
        super().__init__()
        self.transformer_block = transformer_block
        self.adapter = AdapterPfeifferBlock(args, args.embedding_dim, args.adapter_down_size, args.adapter_dropout_rate)
        self.LN = nn.LayerNorm(args.embedding_dim, eps=1e-06)


class SASRecCompacterAdaptedSelfOutput(_Container):  # model.py:650-693
    placement = 'serial'

    def __init__(self, transformer_block, args):
        super().__init__()
        self.transformer_block = transformer_block
        self.adapter1 = HyperComplexAdapterBlock(args, args.embedding_dim, args.adapter_down_size)
        self.adapter2 = HyperComplexAdapterBlock(args, args.embedding_dim, args.adapter_down_size)
