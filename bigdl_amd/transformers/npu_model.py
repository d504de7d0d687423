"""``ipex_llm.transformers.npu_model`` import path on the MI355X.

The reference keeps a second family of ``AutoModel*`` classes for Intel NPUs
(python/llm/src/ipex_llm/transformers/npu_model.py:67-263 ``from_pretrained``, :266-366 ``optimize_npu_model``,
:388-718 ``load_low_bit``): round-to-nearest qtypes (``sym_int4_rtn`` ...), a compile step that fixes
``max_context_len`` / ``max_prompt_len``, multi-process decoder pipelines (``inter_pp`` / ``intra_pp``) and a
``save_directory`` that receives the converted model.  None of the NPU runtime exists here; what a script written
against that module needs is the same call surface.  These classes take the NPU keyword set, translate it to the
GPU loader of ``bigdl_amd.transformers.model`` and place the model on the GPU:

* ``*_rtn`` qtypes map to the block formats of the same width (32-weight blocks are finer than any group size the NPU
  path offers, so ``quantization_group_size`` needs no separate handling);
* ``max_context_len`` / ``max_prompt_len`` are kept on the model (``model.max_context_len``) -- the KV cache here grows
  to whatever the request needs, so nothing is compiled against them;
* ``inter_pp`` / ``intra_pp`` / ``transpose_value_cache`` / ``fuse_layers`` / ``keep_ir`` / ``compile_blob`` /
  ``pipeline`` describe the NPU compiler's partitioning and are accepted and ignored;
* ``save_directory``: the converted model is written there with ``save_low_bit`` (and ``load_low_bit`` of this module
  reads it back);
* ``device="cpu"`` (the reference's mock backend, npu_model.py:138, 207-212) keeps the model on the host.
"""
from __future__ import annotations

import torch

from . import model as _gpu

_RTN = {"sym_int4_rtn": "sym_int4", "asym_int4_rtn": "asym_int4", "sym_int8_rtn": "sym_int8"}
_NPU_ONLY = ("pipeline", "inter_pp", "intra_pp", "transpose_value_cache", "fuse_layers", "keep_ir", "compile_blob",
             "convert_model", "quantization_group_size", "skip_npu_logic")


def _translate(kwargs, default_qtype="sym_int4"):
    for key in _NPU_ONLY:
        kwargs.pop(key, None)
    info = {
        "max_context_len": kwargs.pop("max_context_len", 1024),
        "max_prompt_len": kwargs.pop("max_prompt_len", 512),
        "save_directory": kwargs.pop("save_directory", None),
        "mock_device": kwargs.pop("device", None),
    }
    if "imatrix_file" in kwargs:
        kwargs["imatrix"] = kwargs.pop("imatrix_file")
    kwargs.pop("device_map", None)
    low_bit = kwargs.pop("load_in_low_bit", default_qtype)
    if kwargs.pop("load_in_4bit", False):
        low_bit = "sym_int4"
    info["low_bit"] = _RTN.get(low_bit, low_bit)
    return info


def _finish(model, info):
    model.max_context_len = info["max_context_len"] - 1          # the reference reserves one position, :130
    model.max_prompt_len = info["max_prompt_len"]
    if info["save_directory"] is not None and hasattr(model, "save_low_bit"):
        model.save_low_bit(info["save_directory"])
    if info["mock_device"] in ("cpu", "dummy") or not torch.cuda.is_available():
        return model
    return model.to("cuda")


class _BaseAutoModelClass:
    HF_MODEL = None       # the GPU loader class of the same name

    @classmethod
    def from_pretrained(cls, *args, **kwargs):
        info = _translate(kwargs)
        kwargs.setdefault("optimize_model", True)
        model = cls.HF_MODEL.from_pretrained(*args, load_in_low_bit=info["low_bit"], **kwargs)
        return _finish(model, info)

    @classmethod
    def load_low_bit(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        info = _translate(kwargs)
        info["save_directory"] = None
        return _finish(cls.HF_MODEL.load_low_bit(pretrained_model_name_or_path, *model_args, **kwargs), info)

    @classmethod
    def optimize_npu_model(cls, *args, **kwargs):
        """Convert an already loaded fp16 / fp32 model in place (reference :266-366: ``model=``, ``qtype=``)."""
        from ..optimize import optimize_model
        model = kwargs.pop("model")
        qtype = kwargs.pop("qtype", "sym_int4_rtn")
        info = _translate(kwargs)
        model = optimize_model(model, low_bit=_RTN.get(qtype, qtype),
                               modules_to_not_convert=kwargs.pop("modules_to_not_convert", None) or None)
        return _finish(model, info)


def _make(name):
    return type(name, (_BaseAutoModelClass,), {"HF_MODEL": getattr(_gpu, name), "__doc__": f"NPU-path name of {name}"})


AutoModelForCausalLM = _make("AutoModelForCausalLM")
AutoModel = _make("AutoModel")
AutoModelForSpeechSeq2Seq = _make("AutoModelForSpeechSeq2Seq")
AutoModelForSeq2SeqLM = _make("AutoModelForSeq2SeqLM")
AutoModelForSequenceClassification = _make("AutoModelForSequenceClassification")
AutoModelForMaskedLM = _make("AutoModelForMaskedLM")
AutoModelForQuestionAnswering = _make("AutoModelForQuestionAnswering")
AutoModelForNextSentencePrediction = _make("AutoModelForNextSentencePrediction")
AutoModelForMultipleChoice = _make("AutoModelForMultipleChoice")
AutoModelForTokenClassification = _make("AutoModelForTokenClassification")


class EmbeddingModel(_BaseAutoModelClass):
    """Sentence-embedding models of the NPU path (reference :814-900) load through ``AutoModel``."""
    HF_MODEL = _gpu.AutoModel
