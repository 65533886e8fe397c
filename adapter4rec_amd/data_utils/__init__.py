from .dataset import BuildTrainDataset, BuildEvalDataset, DeviceTrainSampler, SequentialDistributedSampler
from .metrics import eval_model, get_item_embeddings
from .preprocess import read_news_bert, get_doc_input_bert, read_behaviors, read_news
from .utils import *  # noqa: F401,F403
