from .model import (Model, ModelCPC, Vit_Encoder, MAE_Encoder, VITAdaptedSelfOutput, VITAdaptedOutput,   # noqa: F401
                    VITCompacterAdaptedSelfOutput, VITCompacterAdaptedOutput, VITAdaptedParallelOutput, SoftPrompt, VITKAdaptedCVModel,
                    SASRecKAdaptedTransformerBlocks)
from .vit import ViTForImageClassification, ViTMAEModel                                                  # noqa: F401
