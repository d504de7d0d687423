"""One-line patch so unmodified HF + PEFT training scripts use the low-bit model / QLoRA classes
(reference: python/llm/src/ipex_llm/llm_patching.py:35-85)."""
import importlib
import sys

bigdl_patched = None  # None or 'Train'
_attrs = []
_standin = False      # a stand-in ``peft`` module was registered by llm_patch


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
        # this library's modules first: importing them pulls in HuggingFace sub-packages, during which transformers
        # can replace its own (lazy) top-level module object; the names are then patched on the final one
        from .transformers import AutoModel, AutoModelForCausalLM
        from .transformers import qlora
        import transformers
        transformers = sys.modules["transformers"]
        replace_attr(transformers, "AutoModelForCausalLM", AutoModelForCausalLM)
        replace_attr(transformers, "LlamaForCausalLM", AutoModelForCausalLM)
        replace_attr(transformers, "AutoModel", AutoModel)
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
        else:
            # PEFT is not installed: ``import peft`` in the unmodified script resolves to a stand-in module that holds
            # this library's implementations of the names such scripts use (same idea as the reference's
            # utils/modules.py insert_fake_module for packages its patched code paths never call into)
            import types
            global _standin
            mod = types.ModuleType("peft")
            mod.__doc__ = "stand-in registered by llm_patch(train=True): LoRA on low-bit base models"
            for name in ("get_peft_model", "prepare_model_for_kbit_training", "LoraConfig", "PeftModel",
                         "PeftModelForCausalLM", "LoraModel"):
                setattr(mod, name, getattr(qlora, name))
            mod.TaskType = types.SimpleNamespace(CAUSAL_LM="CAUSAL_LM", SEQ_2_SEQ_LM="SEQ_2_SEQ_LM", SEQ_CLS="SEQ_CLS")
            sys.modules["peft"] = mod
            _standin = True
        replace_attr(transformers, "TrainingArguments", qlora.TrainingArguments)
        bigdl_patched = "Train"


def llm_unpatch():
    global bigdl_patched
    if bigdl_patched is None:
        return
    for obj, name, torch_attr in _attrs:
        setattr(obj, name, torch_attr)
    _attrs.clear()
    global _standin
    if _standin:
        sys.modules.pop("peft", None)
        _standin = False
    bigdl_patched = None
