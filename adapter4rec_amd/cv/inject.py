"""Adapter injection for the image path, flag-for-flag as Downstream/CV/run_adapter.py:353-460, and its optimiser groups
(:491-517: LoRA tensors of the ViT carry no 'adapter' in their names and therefore train at --fine_tune_lr)."""
from ..model.lora import LoRALinear
from ..model.model import (CompacterModel, SASRecAdaptedSelfOutput, SASRecCompacterAdaptedSelfOutput,
                           SASRecPfeifferVer2AdaptedSelfOutput)
from ..model.model import SASRecParallelAdaptedSelfOutput
from ..model.model import SASRecKAdaptedTransformerBlocks
from .model import (VITAdaptedOutput, VITAdaptedParallelOutput, VITAdaptedSelfOutput, VITCompacterAdaptedOutput,
                    VITCompacterAdaptedSelfOutput, VITKAdaptedCVModel)


def vit_layers(model):
    net = model.cv_encoder.image_net
    return net.vit.encoder.layer if hasattr(net, 'vit') else net.encoder.layer        # ViTMAEModel has no .vit (run_adapter.py:440-447)


def inject_adapters(model, args):
    if 'None' in args.adding_adapter_to:
        return model
    layers = vit_layers(model)
    blocks = model.user_encoder.transformer_encoder.transformer_blocks
    t = args.adapter_type
    if 'pfeiffer_ver2' in t:                         # :369-378
        for lyr in layers:
            lyr.attention.output = VITAdaptedSelfOutput(lyr.attention.output, args)
        for i, blk in enumerate(blocks):
            blocks[i] = SASRecPfeifferVer2AdaptedSelfOutput(blk, args)
    elif 'kadapter' in t:                            # :378-383 (the reference addresses image_net.vit only: ViT-MAE has no .vit)
        net = model.cv_encoder.image_net
        if not hasattr(net, 'vit'):
            raise NotImplementedError('K-Adapter on ViT-MAE (the reference wires it for ViTForImageClassification only)')
        net.vit.encoder = VITKAdaptedCVModel(net.vit.encoder, args)
        te = model.user_encoder.transformer_encoder
        te.transformer_blocks = SASRecKAdaptedTransformerBlocks(te.transformer_blocks, args)
    elif 'lora' in t:                                # :384-395 (reference hard-codes r = 12 / 4 / 0; BASELINE config 3 asks r = 8)
        r_vit = int(getattr(args, 'lora_r', 12))
        r_q = int(getattr(args, 'lora_r_sasrec', 4))
        for lyr in layers:
            h = lyr.attention.attention.query.in_features          # 768 in the reference (ViT-B)
            lyr.attention.attention.query = LoRALinear(h, h, r=r_vit)
            lyr.attention.attention.value = LoRALinear(h, h, r=r_vit)
        for blk in blocks:
            blk.multi_head_attention.w_Q = LoRALinear(args.embedding_dim, args.embedding_dim, r=r_q)
            blk.multi_head_attention.w_V = LoRALinear(args.embedding_dim, args.embedding_dim)      # r = 0: a plain trainable Linear
    elif 'compacter' in t:                           # :396-412
        for lyr in layers:
            lyr.attention.output = VITCompacterAdaptedSelfOutput(lyr.attention.output, args)
            lyr.output = VITCompacterAdaptedOutput(lyr.output, args)
        for i, blk in enumerate(blocks):
            blocks[i] = SASRecCompacterAdaptedSelfOutput(blk, args)
        model = CompacterModel(args, model)
    elif 'prompt' in t:                              # :413-422: shallow prompt on the ViT + trainable classifier; SASRec untouched
        from .model import SoftPrompt
        net = model.cv_encoder.image_net
        if not hasattr(net, 'vit'):
            raise NotImplementedError('soft prompt on ViT-MAE (the reference wires it for ViTForImageClassification only)')
        h = net.vit.embeddings.cls_token.shape[-1]
        net.vit.embeddings = SoftPrompt(net.vit.embeddings, n_tokens=args.n_tokens, embed_dim=h)
        for name, p in model.named_parameters():
            if 'cv_encoder.image_net.classifier' in name:
                p.requires_grad = True
    elif 'houslby' in t and 'None' not in args.is_serial:      # :425-447
        for lyr in layers:
            lyr.attention.output = VITAdaptedSelfOutput(lyr.attention.output, args)
            lyr.output = VITAdaptedOutput(lyr.output, args)
        for i, blk in enumerate(blocks):
            blocks[i] = SASRecAdaptedSelfOutput(blk, args)
    elif 'houslby' in t:                             # :448-460 --is_serial None: parallel adapter at layer.output only
        for lyr in layers:
            lyr.output = VITAdaptedParallelOutput(lyr.output, args)
        for i, blk in enumerate(blocks):
            blocks[i] = SASRecParallelAdaptedSelfOutput(blk, args)
    else:
        raise NotImplementedError(f'--adapter_type {t} on the image tower')
    getattr(model, 'model', model).invalidate_native()
    return model


def optimizer_groups(model, args):
    groups = dict(img=[], rec=[], acv=[], arec=[])
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        ad = 'adapter' in name
        if 'image_net' in name and not ('fc' in name or 'classifier' in name or 'decoder_pred' in name):
            groups['acv' if ad else 'img'].append(p)
        else:
            groups['arec' if ad else 'rec'].append(p)
    return [{'params': groups['img'], 'lr': args.fine_tune_lr}, {'params': groups['rec'], 'lr': args.lr},
            {'params': groups['acv'], 'lr': args.adapter_cv_lr}, {'params': groups['arec'], 'lr': args.adapter_sasrec_lr}]
