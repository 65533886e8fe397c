"""Host-side readers with the reference's function names, arguments and return tuples
(Downstream/Text/data_utils/preprocess.py).  Pure Python / numpy: runs before the hot path."""
import numpy as np
import torch


def read_news(news_path):
    """preprocess.py:66-77: '<doc_name>\\t<text>' per line -> ({id: name}, {name: id}), ids from 1."""
    id_to_name, name_to_id = {}, {}
    with open(news_path, 'r') as f:
        for i, line in enumerate(f, start=1):
            name, _ = line.strip('\n').split('\t')
            name_to_id[name] = i
            id_to_name[i] = name
    return id_to_name, name_to_id


def read_news_bert(news_path, args, tokenizer):
    """preprocess.py:80-107: tokenise the lower-cased title to exactly num_words_title ids (+ attention mask).
    abstract / body (--news_attributes, :93-103): the reference reads them from variables its two-column split never defines (a NameError as
    shipped); here they are the optional 3rd / 4th tab-separated columns of the news file, tokenised to num_words_abstract / num_words_body
    (the body cut at 2 000 characters first, :100)."""
    item_id_to_dic, item_name_to_id = {}, {}
    want = set(args.news_attributes)
    with open(news_path, 'r') as f:
        for item_id, line in enumerate(f, start=1):
            cols = line.strip('\n').split('\t')
            doc_name, title = cols[0], cols[1]
            tok = tokenizer(title.lower(), max_length=args.num_words_title, padding='max_length', truncation=True) if 'title' in want else []
            extra = []
            for k, (name, nw, cut) in enumerate((('abstract', getattr(args, 'num_words_abstract', 0), None), ('body', getattr(args, 'num_words_body', 0), 2000))):
                if name in want:
                    if len(cols) < 3 + k:
                        raise ValueError(f'--news_attributes {name}: {news_path} has no column {3 + k} (doc_name, title, abstract, body)')
                    extra.append(tokenizer(cols[2 + k].lower()[:cut], max_length=nw, padding='max_length', truncation=True))
                else:
                    extra.append([])
            item_name_to_id[doc_name] = item_id
            item_id_to_dic[item_id] = [tok] + extra
    return item_id_to_dic, item_name_to_id


def get_doc_input_bert(item_id_to_content, args):
    """preprocess.py:110-151: int32 [item_num + 1, L] id and mask matrices per attribute (row 0 = PAD item); None for the attributes not asked for."""
    n = len(item_id_to_content) + 1
    out = []
    for k, (name, nw) in enumerate((('title', args.num_words_title), ('abstract', getattr(args, 'num_words_abstract', 0)), ('body', getattr(args, 'num_words_body', 0)))):
        if name not in args.news_attributes:
            out += [None, None]
            continue
        ids, mask = np.zeros((n, nw), dtype='int32'), np.zeros((n, nw), dtype='int32')
        for item_id in range(1, n):
            t = item_id_to_content[item_id][k]
            ids[item_id] = t['input_ids']
            mask[item_id] = t['attention_mask']
        out += [ids, mask]
    return tuple(out)


def read_behaviors(behaviors_path, before_item_id_to_dic, before_item_name_to_id, max_seq_len, min_seq_len, Log_file):
    """preprocess.py:5-63.  Keeps the last max_seq_len + 3 interactions of every user with >= min_seq_len of them,
    re-numbers the items that survive from 1, and splits each sequence:
      train = seq[:-2], valid = seq[-(max_seq_len + 2):-1], test = seq[-(max_seq_len + 1):]
      history for valid = train, history for test = seq[:-1]."""
    n_before = len(before_item_name_to_id)
    Log_file.info('##### news number {} {} (before clearing)#####'.format(len(before_item_id_to_dic), n_before))
    Log_file.info('##### min seq len {}, max seq len {}#####'.format(min_seq_len, max_seq_len))
    counts = [0] * (n_before + 1)
    user_seqs = {}
    with open(behaviors_path, 'r') as f:
        for line in f:
            parts = line.strip('\n').split('\t')
            names = parts[1].split(' ')
            if len(names) < min_seq_len:
                continue
            ids = [before_item_name_to_id[x] for x in names[-(max_seq_len + 3):]]
            user_seqs[parts[0]] = ids
            for i in ids:
                counts[i] += 1
    remap, item_id_to_dic = {}, {}
    nxt = 1
    for old in range(1, n_before + 1):
        if counts[old]:
            remap[old] = nxt
            item_id_to_dic[nxt] = before_item_id_to_dic[old]
            nxt += 1
    item_num = len(remap)
    Log_file.info('##### items after clearing {}, {}, {} #####'.format(item_num, len(remap), len(item_id_to_dic)))
    train, valid, test, hist_valid, hist_test = {}, {}, {}, {}, {}
    for uid, seq in enumerate(user_seqs.values()):
        s = [remap[i] for i in seq]
        train[uid] = s[:-2]
        valid[uid] = s[-(max_seq_len + 2):-1]
        test[uid] = s[-(max_seq_len + 1):]
        hist_valid[uid] = torch.LongTensor(np.array(s[:-2]))
        hist_test[uid] = torch.LongTensor(np.array(s[:-1]))
    Log_file.info('##### user seqs after clearing {}, {}, {}, {}, {}#####'.format(
        len(user_seqs), len(user_seqs), len(train), len(valid), len(test)))
    return item_num, item_id_to_dic, train, valid, test, hist_valid, hist_test
