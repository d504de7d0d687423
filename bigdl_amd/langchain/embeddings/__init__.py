from .bigdlllm import (BigdlNativeEmbeddings, BloomEmbeddings, GptneoxEmbeddings, LlamaEmbeddings,  # noqa: F401
                       StarcoderEmbeddings)
from .transformersembeddings import TransformersBgeEmbeddings, TransformersEmbeddings  # noqa: F401
