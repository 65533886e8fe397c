"""Every flag of Downstream/CV/parameters.py:8-67, same spellings and defaults (+ --compute_dtype, --lora_r, --lora_r_sasrec,
--local-rank)."""
import argparse


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--mode', type=str, default='train')
    p.add_argument('--item_tower', type=str, default='modal')
    p.add_argument('--root_data_dir', type=str, default='../')
    p.add_argument('--dataset', type=str, default='pinterest')
    p.add_argument('--behaviors', type=str, default='users_log.tsv')
    p.add_argument('--images', type=str, default='images_log.tsv')
    p.add_argument('--lmdb_data', type=str, default='image.lmdb')
    p.add_argument('--batch_size', type=int, default=64)
    p.add_argument('--epoch', type=int, default=1)
    p.add_argument('--lr', type=float, default=1e-3)
    p.add_argument('--fine_tune_lr', type=float, default=1e-5)
    p.add_argument('--l2_weight', type=float, default=0)
    p.add_argument('--drop_rate', type=float, default=0.1)
    p.add_argument('--CV_model_load', type=str, default='resnet-50')
    p.add_argument('--freeze_paras_before', type=int, default=45)
    p.add_argument('--CV_resize', type=int, default=224)
    p.add_argument('--embedding_dim', type=int, default=64)
    p.add_argument('--num_attention_heads', type=int, default=2)
    p.add_argument('--transformer_block', type=int, default=2)
    p.add_argument('--max_seq_len', type=int, default=10)
    p.add_argument('--min_seq_len', type=int, default=5)
    p.add_argument('--arch', type=str, default='sasrec')
    p.add_argument('--use_scale', type=str, default='half')       # reference: fp16 autocast; here the engine's bf16 storage / fp32 accumulate
    p.add_argument('--n_tokens', type=int, default=10)
    p.add_argument('--num_workers', type=int, default=12)
    p.add_argument('--load_ckpt_name', type=str, default='None')
    p.add_argument('--label_screen', type=str, default='None')
    p.add_argument('--logging_num', type=int, default=8)
    p.add_argument('--testing_num', type=int, default=1)
    p.add_argument('--local_rank', '--local-rank', default=-1, type=int)
    p.add_argument('--pretrained_recsys_model', default='None', type=str)
    p.add_argument('--adapter_down_size', type=int, default=16)
    p.add_argument('--adding_adapter_to', type=str, default='bert')
    p.add_argument('--fine_tune_to', type=str, default='None')
    p.add_argument('--adapter_cv_lr', type=float, default=5e-4)
    p.add_argument('--adapter_sasrec_lr', type=float, default=1e-4)
    p.add_argument('--cv_adapter_down_size', type=int, default=64)
    p.add_argument('--adapter_dropout_rate', type=float, default=0.1)
    p.add_argument('--adapter_activation', type=str, default='RELU')
    p.add_argument('--finetune_layernorm', type=str, default='None')
    p.add_argument('--is_serial', type=str, default='True')
    p.add_argument('--adapter_type', type=str, default='houslby')
    p.add_argument('--k_adapter_bert_list', type=str, default='0,11')
    p.add_argument('--k_adapter_bert_hidden_dim', type=int, default=384)
    p.add_argument('--num_adapter_heads_sasrec', type=int, default=2)
    p.add_argument('--num_adapter_heads_bert', type=int, default=12)
    p.add_argument('--num_dnn', type=int, default=0)
    p.add_argument('--hypercomplex_division', type=int, default=4)
    p.add_argument('--phm_init_range', type=float, default=0.0001)
    p.add_argument('--compute_dtype', type=str, default=None, choices=[None, 'bf16', 'fp32', 'fp8'],
                   help="fp8: bf16 storage + OCP e4m3 operands for the frozen backbone's qkv / FFN-up forward GEMMs (BASELINE config 5)")
    p.add_argument('--residual_dtype', type=str, default='bf16', choices=['bf16', 'fp32'],
                   help='accepted for symmetry with the text entry point; fp32 raises on the image tower (pre-LN: the residual stream is the stored tensor v)')
    p.add_argument('--eval_compute_dtype', type=str, default=None, choices=[None, 'bf16', 'fp32'],
                   help="dtype of eval's item sweep; None = --compute_dtype (the reference evaluates under the same AMP setting it trains with)")
    p.add_argument('--lora_r', type=int, default=12)              # run_adapter.py:386-387 hard-codes 12
    p.add_argument('--lora_r_sasrec', type=int, default=4)        # :393 hard-codes 4
    args = p.parse_args(argv)
    if args.compute_dtype is None:
        args.compute_dtype = 'bf16' if 'half' in args.use_scale else 'fp32'
    return args
