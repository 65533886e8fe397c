from .bert import BertBackbone, BERT_BASE, ROBERTA_BASE
from .encoders import Bert_Encoder, Text_Encoder, User_Encoder
from .model import (Model, ModelCPC, CompacterModel, BertAdaptedSelfOutput, BertAdaptedParallelSelfOutput,
                    BertPfeifferAdaptedSelfOutput, BertCompacterAdaptedSelfOutput, SASRecAdaptedSelfOutput,
                    SASRecParallelAdaptedSelfOutput, SASRecPfeifferAdaptedSelfOutput,
                    SASRecPfeifferVer2AdaptedSelfOutput, SASRecCompacterAdaptedSelfOutput, SoftEmbedding,
                    BertKAdaptedBertModel, SASRecKAdaptedTransformerBlocks)
from .modules import (AdapterBlock, AdapterPfeifferBlock, HyperComplexAdapterBlock, KAdapterBlock, PHMLinear, TransformerBlock,
                      TransformerEncoder, MultiHeadedAttention, PositionwiseFeedForward)
from .lora import LoRALinear
