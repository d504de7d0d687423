"""``BigdlNativeForCausalLM`` and the per-family ``*ForCausalLM(native=True)`` loaders of single-file models.

Reference: python/llm/src/ipex_llm/transformers/modelling_bigdl.py:29-157 (dispatch on ``model_family`` to the ctypes
wrappers ``Llama`` / ``Gptneox`` / ``Bloom`` / ``Starcoder`` / ``ChatGLM``).
"""
from __future__ import annotations

from ..utils.common import invalidInputError


class BigdlNativeForCausalLM:
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, model_family: str = "llama", **kwargs):
        invalidInputError(model_family in ("llama", "gptneox", "bloom", "starcoder", "chatglm"),
                          "Now we only support model family: 'llama', 'gptneox', 'bloom', 'starcoder', 'chatglm', "
                          f"'{model_family}' is not in the list.")
        from ..ggml import model as M
        klass = {"llama": M.Llama, "gptneox": M.Gptneox, "bloom": M.Bloom, "starcoder": M.Starcoder,
                 "chatglm": M.ChatGLM}[model_family]
        return klass(model_path=pretrained_model_name_or_path, **kwargs)


class _BaseGGMLClass:
    GGML_Model = None
    HF_Class = None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, native: bool = True, dtype: str = "int4", *args, **kwargs):
        if native:
            from ..ggml import model as M
            return getattr(M, cls.GGML_Model)(model_path=pretrained_model_name_or_path, **kwargs)
        from . import AutoModelForCausalLM
        return AutoModelForCausalLM.from_pretrained(pretrained_model_name_or_path, *args,
                                                    load_in_low_bit={"int4": "sym_int4", "int8": "sym_int8"}.get(dtype, dtype),
                                                    **kwargs)


class LlamaForCausalLM(_BaseGGMLClass):
    GGML_Model = "Llama"


class ChatGLMForCausalLM(_BaseGGMLClass):
    GGML_Model = "ChatGLM"


class GptneoxForCausalLM(_BaseGGMLClass):
    GGML_Model = "Gptneox"


class BloomForCausalLM(_BaseGGMLClass):
    GGML_Model = "Bloom"


class StarcoderForCausalLM(_BaseGGMLClass):
    GGML_Model = "Starcoder"
