"""Bert_Encoder / Text_Encoder / User_Encoder with the reference's interface
(Downstream/Text/model/encoders.py): callable entry points that run on the native engine."""
import torch
from torch import nn
from torch.nn.init import constant_, xavier_normal_

from .bert import _Container
from .modules import TransformerEncoder


class User_Encoder(nn.Module):                  # encoders.py:8-29
    def __init__(self, item_num, max_seq_len, item_dim, num_attention_heads, dropout, n_layers):
        super().__init__()
        self.transformer_encoder = TransformerEncoder(n_vocab=item_num, n_position=max_seq_len, d_model=item_dim,
                                                      n_heads=num_attention_heads, dropout=dropout, n_layers=n_layers)
        for m in self.modules():                # reference _init_weights: xavier-normal Linear/Embedding, zero bias
            if isinstance(m, (nn.Embedding, nn.Linear)):
                xavier_normal_(m.weight.data)
                if isinstance(m, nn.Linear) and m.bias is not None:
                    constant_(m.bias.data, 0)
        self._owner = [None]                    # list: keeps the owning Model out of the module tree

    def forward(self, input_embs, log_mask, local_rank=None):
        """[b, T, E] item embeddings + [b, T] log_mask -> [b, T, E] (inference path used by eval, metrics.py:101-104)."""
        return self._owner[0]._engine().user_encode(input_embs, log_mask)


class Text_Encoder(_Container):                 # encoders.py:38-57
    def __init__(self, bert_model, item_embedding_dim, word_embedding_dim):
        super().__init__()
        self.bert_model = bert_model
        self.fc = nn.Linear(word_embedding_dim, item_embedding_dim)
        self.activate = nn.GELU()


class Bert_Encoder(nn.Module):                  # encoders.py:60-99
    def __init__(self, args, bert_model):
        super().__init__()
        self.args = args
        assert len(args.news_attributes) > 0
        lengths = {'title': args.num_words_title * 2, 'abstract': args.num_words_abstract * 2, 'body': args.num_words_body * 2}
        for k in lengths:
            if k not in args.news_attributes:
                lengths[k] = 0
        self.attributes2length = lengths
        self.attributes2start, s = {}, 0
        for k in ('title', 'abstract', 'body'):
            self.attributes2start[k] = s
            s += lengths[k]
        self.text_encoders = nn.ModuleDict({'title': Text_Encoder(bert_model, args.embedding_dim, args.word_embedding_dim)})
        # every attribute goes through the ONE Text_Encoder the reference builds ('title') and the item vector is the mean (encoders.py:89-99).  The
        # native engine stacks the attributes as extra items at the longest attribute's length (engine.py: _stack_attrs), in this canonical order:
        self.newsname = [n for n in ('title', 'abstract', 'body') if n in set(args.news_attributes)]
        if not self.newsname:
            raise ValueError(f'--news_attributes {args.news_attributes}: none of title, abstract, body')
        self._owner = [None]

    def forward(self, news):
        """[n, 2*num_words] ids||mask -> [n, E] item embeddings (inference; eval uses it, metrics.py:62-79)."""
        return self._owner[0]._engine().encode_items(news)
