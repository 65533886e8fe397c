"""One tiny adapter-tuned TransRec training step on cuda:0 through the HIP kernels, checked against the CPU oracle
(oracle/ref_cpu.py; the oracle is only the checker here)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_smoke():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from golden_util import load_variant, strip, LRS
    from oracle import ref_cpu as R
    from adapter4rec_amd.inject import freeze_all, inject_adapters
    from adapter4rec_amd.model import BertBackbone, Model

    sd, cfg, fx, trainable, (items, mask), _ = load_variant('houlsby')
    args = argparse.Namespace(
        max_seq_len=20, l2_weight=0, embedding_dim=64, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
        num_words_title=30, num_words_abstract=50, num_words_body=50, news_attributes=['title'], word_embedding_dim=128,
        bert_model_load='bert_tiny', bert_adapter_down_size=64, adapter_down_size=16, adapter_dropout_rate=0.1,
        adapter_activation='RELU', hypercomplex_division=4, phm_init_range=1e-4, adapter_type='houslby', is_serial='True',
        adding_adapter_to='all', arch='sasrec', compute_dtype='fp32', **LRS)
    geom = dict(vocab_size=120, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                max_position_embeddings=40, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
                attention_probs_dropout_prob=0.1, pad_token_id=0, model_type='bert')
    model = Model(args, 200, True, BertBackbone(geom))
    freeze_all(model)
    model = inject_adapters(model, args)
    model.load_state_dict({str(k): sd[strip(str(k))] for k in fx['all_keys']}, strict=True)
    model.to('cuda:0').eval()
    loss = model(items.to('cuda:0'), mask.to('cuda:0'), 0)
    loss.backward()
    torch.cuda.synchronize()
    out, grads = R.loss_and_grads(sd, trainable, items, mask, cfg)
    assert abs(loss.item() - float(out['loss'])) < 1e-4, (loss.item(), float(out['loss']))
    params = dict(model.named_parameters())
    worst = 0.0
    for k in trainable:
        ref = grads[k].numpy()
        got = params[k].grad.cpu().numpy()
        worst = max(worst, float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)))
    assert worst < 1e-3, worst
    print(f'smoke ok: loss {loss.item():.6f} (oracle {float(out["loss"]):.6f}), worst adapter-grad rel err {worst:.2e}')
