"""The committed heads of the reference's shipped data files (tests/golden/real_data/, written by tools/gen_golden_r6.py: data, not source):
the first 1 024 lines of Dataset/Adressa/Adressa_news_base.tsv and of Dataset/Amazon/amazon_2w_users.tsv, the whole Amazon item list and the
bert_base_uncased vocabulary (both gzipped).  `head_paths()` unpacks them into a temporary directory shaped like what the readers and
BertTokenizer.from_pretrained expect."""
import contextlib
import gzip
import os
import shutil
import tempfile

REAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'real_data')


@contextlib.contextmanager
def head_paths():
    d = tempfile.mkdtemp(prefix='a4r_real_')
    try:
        vdir = os.path.join(d, 'bert_base_uncased')
        os.makedirs(vdir)
        with gzip.open(os.path.join(REAL, 'bert_base_uncased_vocab.txt.gz'), 'rb') as f, open(os.path.join(vdir, 'vocab.txt'), 'wb') as g:
            shutil.copyfileobj(f, g)
        with gzip.open(os.path.join(REAL, 'amazon_items.tsv.gz'), 'rb') as f, open(os.path.join(d, 'amazon_items.tsv'), 'wb') as g:
            shutil.copyfileobj(f, g)
        yield dict(vocab_dir=vdir, news=os.path.join(REAL, 'adressa_news_head.tsv'), users=os.path.join(REAL, 'amazon_users_head.tsv'),
                   items=os.path.join(d, 'amazon_items.tsv'))
    finally:
        shutil.rmtree(d, ignore_errors=True)
