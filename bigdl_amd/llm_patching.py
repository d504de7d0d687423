"""One-line patch so unmodified HF + PEFT training scripts use the low-bit model / QLoRA classes
(reference: python/llm/src/ipex_llm/llm_patching.py:35-85)."""
import importlib
import sys

bigdl_patched = None  # None or 'Train'
_attrs = []


def replace_attr(obj, name: str, value):
    original_attr = getattr(obj, name)
    setattr(obj, name, value)
    _attrs.append((obj, name, original_attr))


def llm_patch(train=False):
    """``llm_patch(train=True)`` before importing peft: ``transformers.AutoModelForCausalLM`` /
    ``LlamaForCausalLM`` / ``AutoModel`` load through this library, and ``peft.get_peft_model`` /
    ``prepare_model_for_kbit_training`` / ``LoraConfig`` / ``TrainingArguments`` resolve to the QLoRA-aware ones."""
    global bigdl_patched
    if bigdl_patched:
        return
    if train:
        import transformers
        from .transformers import AutoModel, AutoModelForCausalLM
        replace_attr(transformers, "AutoModelForCausalLM", AutoModelForCausalLM)
        replace_attr(transformers, "LlamaForCausalLM", AutoModelForCausalLM)
        replace_attr(transformers, "AutoModel", AutoModel)
        from .transformers import qlora
        peft = sys.modules.get("peft")
        if peft is None:
            try:
                peft = importlib.import_module("peft")
            except ImportError:
                peft = None
        if peft is not None:
            replace_attr(peft, "get_peft_model", qlora.get_peft_model)
            replace_attr(peft, "prepare_model_for_kbit_training", qlora.prepare_model_for_kbit_training)
            replace_attr(peft, "LoraConfig", qlora.LoraConfig)
        replace_attr(transformers, "TrainingArguments", qlora.TrainingArguments)
        bigdl_patched = "Train"


def llm_unpatch():
    global bigdl_patched
    if bigdl_patched is None:
        return
    for obj, name, torch_attr in _attrs:
        setattr(obj, name, torch_attr)
    _attrs.clear()
    bigdl_patched = None
