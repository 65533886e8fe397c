"""Adapter injection by module replacement, flag-for-flag as Downstream/Text/run.py:385-479
(substring tests on --adapter_type incl. the reference's spelling 'houslby', 'pfeiffer_ver2' tested
before 'pfeiffer', --is_serial 'None' selecting the parallel form)."""
from .model import (BertAdaptedParallelSelfOutput, BertAdaptedSelfOutput, BertCompacterAdaptedSelfOutput,
                    BertPfeifferAdaptedSelfOutput, CompacterModel, SASRecAdaptedSelfOutput,
                    SASRecCompacterAdaptedSelfOutput, SASRecParallelAdaptedSelfOutput,
                    SASRecPfeifferAdaptedSelfOutput, SASRecPfeifferVer2AdaptedSelfOutput)


def freeze_all(model):
    """--fine_tune_to None (run.py:369-371)."""
    for p in model.parameters():
        p.requires_grad = False


def inject_adapters(model, args):
    """Returns the (possibly wrapped: CompacterModel) model with adapters attached."""
    if 'None' in args.adding_adapter_to:
        return model
    t = args.adapter_type
    layers = model.bert_encoder.text_encoders['title'].bert_model.encoder.layer
    blocks = model.user_encoder.transformer_encoder.transformer_blocks
    if 'pfeiffer_ver2' in t:
        for lyr in layers:
            lyr.attention.output = BertAdaptedSelfOutput(lyr.attention.output, args)
        for i, blk in enumerate(blocks):
            blocks[i] = SASRecPfeifferVer2AdaptedSelfOutput(blk, args)
    elif 'pfeiffer' in t:
        for lyr in layers:
            lyr.output = BertPfeifferAdaptedSelfOutput(lyr.output, args)
        for i, blk in enumerate(blocks):
            blocks[i] = SASRecPfeifferAdaptedSelfOutput(blk, args)
    elif 'kadapter' in t:                               # run.py:409-413
        from .model.model import BertKAdaptedBertModel, SASRecKAdaptedTransformerBlocks
        te = model.bert_encoder.text_encoders['title']
        te.bert_model = BertKAdaptedBertModel(te.bert_model, args)
        ue = model.user_encoder.transformer_encoder
        ue.transformer_blocks = SASRecKAdaptedTransformerBlocks(ue.transformer_blocks, args)
    elif 'prompt' in t:                                 # run.py:429-434
        from .model.model import SoftEmbedding
        bm = model.bert_encoder.text_encoders['title'].bert_model
        bm.set_input_embeddings(SoftEmbedding(bm.get_input_embeddings(), n_tokens=args.n_tokens, initialize_from_vocab=True))
    elif 'lora' in t:                                   # run.py:414-428: fresh lora.Linear modules replace q, v / w_Q, w_V
        from .model.lora import LoRALinear
        h = model.bert_encoder.text_encoders['title'].fc.in_features
        for lyr in layers:
            lyr.attention.self.query = LoRALinear(h, h, r=args.bert_adapter_down_size)
            lyr.attention.self.value = LoRALinear(h, h, r=args.bert_adapter_down_size)
        for blk in blocks:
            blk.multi_head_attention.w_Q = LoRALinear(args.embedding_dim, args.embedding_dim, r=args.adapter_down_size)
            blk.multi_head_attention.w_V = LoRALinear(args.embedding_dim, args.embedding_dim, r=args.adapter_down_size)
    elif 'compacter' in t:
        for lyr in layers:
            lyr.attention.output = BertCompacterAdaptedSelfOutput(lyr.attention.output, args)
            lyr.output = BertCompacterAdaptedSelfOutput(lyr.output, args)
        for i, blk in enumerate(blocks):
            blocks[i] = SASRecCompacterAdaptedSelfOutput(blk, args)
        model = CompacterModel(args, model)
    elif 'houslby' in t:
        serial = 'None' not in args.is_serial
        bw = BertAdaptedSelfOutput if serial else BertAdaptedParallelSelfOutput
        sw = SASRecAdaptedSelfOutput if serial else SASRecParallelAdaptedSelfOutput
        for lyr in layers:
            lyr.attention.output = bw(lyr.attention.output, args)
            lyr.output = bw(lyr.output, args)
        for i, blk in enumerate(blocks):
            blocks[i] = sw(blk, args)
    inner = getattr(model, 'model', model)
    inner.invalidate_native()
    return model


def optimizer_groups(model, args):
    """The four lr groups of run.py:505-529 (substring tests on parameter names)."""
    groups = dict(bert=[], rec=[], abert=[], arec=[])
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        ad = 'adapter' in name or 'lora' in name
        if 'bert_encoder' in name:
            groups['abert' if ad else 'bert'].append(p)
        else:
            groups['arec' if ad else 'rec'].append(p)
    out = [{'params': groups['bert'], 'lr': args.fine_tune_lr}, {'params': groups['rec'], 'lr': args.lr},
           {'params': groups['abert'], 'lr': args.adapter_bert_lr}, {'params': groups['arec'], 'lr': args.adapter_sasrec_lr}]
    return out
