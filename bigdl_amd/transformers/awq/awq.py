"""Swap nn.Linear for AWQ containers before loading an AWQ checkpoint
(reference: python/llm/src/ipex_llm/transformers/awq/awq.py)."""
import torch.nn as nn

from .awq_config import AWQLinearVersion
from .linear import WQLinear_GEMM, WQLinear_GEMV

layer_type_dict = {
    "mpt": "MPTBlock", "llama": "LlamaDecoderLayer", "opt": "OPTDecoderLayer", "RefinedWeb": "FalconDecoderLayer",
    "falcon": "FalconDecoderLayer", "bloom": "BloomBlock", "gptj": "GPTJBlock", "gpt_bigcode": "GPTBigCodeBlock",
    "mistral": "MistralDecoderLayer", "gpt_neox": "GPTNeoXDecoderLayer", "aquila": "AquilaDecoderLayer",
    "Yi": "YiDecoderLayer", "mixtral": "MixtralDecoderLayer", "qwen2": "Qwen2DecoderLayer",
}


def get_layer_type(config):
    return layer_type_dict.get(config.model_type)


def get_blocks(model):
    for path in ("model.layers", "transformer.h", "transformer.blocks", "gpt_neox.layers", "model.decoder.layers"):
        obj = model
        try:
            for p in path.split("."):
                obj = getattr(obj, p)
            return obj
        except AttributeError:
            continue
    raise NotImplementedError(type(model))


def _replace_with_awq_layers(model, awq_config, modules_to_not_convert=None, prefix=""):
    modules_to_not_convert = modules_to_not_convert or ["lm_head"]
    cls = WQLinear_GEMM if awq_config.version == AWQLinearVersion.GEMM else WQLinear_GEMV
    backend = awq_config.backend.value if hasattr(awq_config.backend, "value") else awq_config.backend
    for name, module in model.named_children():
        full = f"{prefix}.{name}" if prefix else name
        if isinstance(module, nn.Linear) and not any(k in full for k in modules_to_not_convert):
            model._modules[name] = cls(awq_config.bits, awq_config.group_size, module.in_features,
                                       module.out_features, module.bias is not None, module.weight.device,
                                       backend=backend)
        elif len(list(module.children())) > 0:
            _replace_with_awq_layers(module, awq_config, modules_to_not_convert, full)
    return model
