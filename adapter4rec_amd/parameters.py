"""Command-line flags of the reference, same spellings and defaults (Downstream/Text/parameters.py:4-86, incl. the
misspelt default --adapter_type houslby), plus: --compute_dtype {bf16,fp32,fp8}, and --local-rank / LOCAL_RANK accepted
next to --local_rank (torch >= 2.0 launchers pass the hyphenated form)."""
import argparse
import os


def build_parser():
    p = argparse.ArgumentParser()
    # ============== data_dir ==============
    p.add_argument('--mode', type=str, default='train', choices=['train', 'test', 'load'])
    p.add_argument('--item_tower', type=str, default='modal', choices=['modal', 'id'])
    p.add_argument('--root_data_dir', type=str, default='../')
    p.add_argument('--dataset', type=str, default='Adressa')
    p.add_argument('--behaviors', type=str, default='Adressa_users_base.tsv')
    p.add_argument('--news', type=str, default='Adressa_news_base.tsv')
    # ============== train parameters ==============
    p.add_argument('--batch_size', type=int, default=64)
    p.add_argument('--epoch', type=int, default=1)
    p.add_argument('--lr', type=float, default=1e-5)
    p.add_argument('--fine_tune_lr', type=float, default=1e-5)
    p.add_argument('--l2_weight', type=float, default=0)
    p.add_argument('--drop_rate', type=float, default=0.1)
    # ============== model parameters ==============
    p.add_argument('--bert_model_load', type=str, default='bert-base-uncased')
    p.add_argument('--freeze_paras_before', type=int, default=165)
    p.add_argument('--word_embedding_dim', type=int, default=768)
    p.add_argument('--embedding_dim', type=int, default=256)
    p.add_argument('--num_attention_heads', type=int, default=2)
    p.add_argument('--transformer_block', type=int, default=2)
    p.add_argument('--max_seq_len', type=int, default=20)
    p.add_argument('--min_seq_len', type=int, default=5)
    p.add_argument('--use_cls', type=bool, default=True)
    # ============== switch and logging setting ==============
    p.add_argument('--num_workers', type=int, default=12)
    p.add_argument('--load_ckpt_name', type=str, default='None')
    p.add_argument('--label_screen', type=str, default='None')
    p.add_argument('--logging_num', type=int, default=8)
    p.add_argument('--testing_num', type=int, default=1)
    p.add_argument('--local_rank', '--local-rank', dest='local_rank', default=int(os.environ.get('LOCAL_RANK', -1)), type=int)
    # ============== news information ==============
    p.add_argument('--num_words_title', type=int, default=30)
    p.add_argument('--num_words_abstract', type=int, default=50)
    p.add_argument('--num_words_body', type=int, default=50)
    p.add_argument('--news_attributes', type=str, default='title')
    # ============== transfer learning ==============
    p.add_argument('--now_epoch', type=int, default=1)
    p.add_argument('--pretrained_model_dir', type=str, default='pretrained_RecSys_model')
    p.add_argument('--pretrained_model_name', type=str, default='epoch-15')
    # ============== adapters ==============
    p.add_argument('--adapter_down_size', type=int, default=16)
    p.add_argument('--adding_adapter_to', type=str, default='bert')
    p.add_argument('--fine_tune_to', type=str, default='None')
    p.add_argument('--adapter_bert_lr', type=float, default=5e-4)
    p.add_argument('--adapter_sasrec_lr', type=float, default=1e-4)
    p.add_argument('--bert_adapter_down_size', type=int, default=64)
    p.add_argument('--adapter_dropout_rate', type=float, default=0.1)
    p.add_argument('--adapter_activation', type=str, default='RELU')
    p.add_argument('--finetune_layernorm', type=str, default='None')
    p.add_argument('--is_serial', type=str, default='True')
    p.add_argument('--adapter_type', type=str, default='houslby')
    p.add_argument('--k_adapter_bert_list', type=str, default='0,11')
    p.add_argument('--k_adapter_bert_hidden_dim', type=int, default=384)
    p.add_argument('--num_adapter_heads_sasrec', type=int, default=2)
    p.add_argument('--num_adapter_heads_bert', type=int, default=12)
    # ============= architecture / prompt / compacter ==================
    p.add_argument('--arch', type=str, default='sasrec')
    p.add_argument('--n_tokens', type=int, default=30)
    p.add_argument('--initialize_from_vocab', type=int, default=True)
    p.add_argument('--is_use_prompt', type=str, default='True')
    p.add_argument('--hypercomplex_division', type=int, default=4)
    p.add_argument('--phm_init_range', type=float, default=0.0001)
    # ============= native path ==================
    p.add_argument('--compute_dtype', type=str, default='bf16', choices=['bf16', 'fp32', 'fp8'],
                   help='item-encoder storage type on the MI355X path (fp32 = reference precision of Downstream/Text)')
    p.add_argument('--device_sampler', type=int, default=0,
                   help='1: training batches are drawn ON THE GPU (data_utils.DeviceTrainSampler: user sequences and item contents resident in HBM, '
                        "negatives by vectorised rejection) instead of BuildTrainDataset + DataLoader workers -- the same distribution as "
                        'Downstream/Text/data_utils/dataset.py:24-49, not the same random stream; the host then only enqueues.  0 (default): the '
                        "reference's DataLoader path (bit-pinned); it keeps up with the GPU from ~4 workers on (profiles/r04_*_run_throughput.json)")
    p.add_argument('--residual_dtype', type=str, default='bf20', choices=['bf16', 'fp32', 'bf24', 'bf20'],
                   help="bf16 storage only: how the item encoder's residual stream between sub-layers is kept.  The reference's autocast(bfloat16) keeps it "
                        'in fp32 (its LayerNorm outputs fp32).  bf20 (default since round 6): the bf16 tensor + a NIBBLE per element (4 more mantissa bits, rounded) '
                        'on the sub-layers that run the one-launch serial adapter kernel (Houlsby / Compacter): scores / embeddings at 0.5 - 0.9x the distance '
                        "of the reference's own autocast path from fp32, for +1.6 %% of the BERT-base + Houlsby step.  bf24: a byte per element (8 more bits; the same "
                        'accuracy, +2.4 %%).  fp32: fp32 twins (0.5 - 0.9x, +5.3 %%; also un-adapted sub-layers and Pfeiffer through a4r_ln_fwd_sum).  bf16: no twin '
                        '(1.2 - 1.3x, the fastest).  Parallel Houlsby and K-Adapter blocks keep the bf16 stream; text tower only')
    p.add_argument('--eval_compute_dtype', type=str, default='fp32', choices=['bf16', 'fp32'],
                   help="dtype of eval's item sweep (get_item_embeddings).  Default fp32 = the reference's eval precision: HR@10 / nDCG@10 "
                        'and per-user ranks then match the fp32 reference exactly on the trained weights (a bf16 sweep moves a few users across '
                        'the rank-10 boundary: ~2e-3 in HR@10); costs ~3 %% of an epoch on MIND-size item sets')
    return p


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    args.news_attributes = args.news_attributes.split(',')
    return args


if __name__ == '__main__':
    print(parse_args())
