"""Glue for HuggingFace ``Trainer`` on this stack.

The reference patches ``accelerate`` / ``TrainingArguments._setup_devices`` so that training runs on XPU with oneCCL
(python/llm/src/ipex_llm/transformers/training_patch.py:68-220).  On ROCm the stock device setup already selects
``cuda:LOCAL_RANK`` and the ``nccl`` (= RCCL) backend, so nothing in ``accelerate`` has to be replaced; what remains
is (a) making ``Trainer`` save adapter-only checkpoints for our own PEFT wrapper and (b) environment defaults.
"""
from __future__ import annotations

import os


def setup_training_env():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("TOKENIZERS_PARALLELISM", "false")


def patch_trainer_save():
    """``Trainer._save`` stores only the adapter when the model is a ``PeftModelForCausalLM``."""
    from transformers import Trainer
    from .qlora import PeftModelForCausalLM
    if getattr(Trainer, "_bigdl_save_patched", False):
        return
    orig = Trainer._save

    def _save(self, output_dir=None, state_dict=None):
        model = self.model
        if isinstance(model, PeftModelForCausalLM):
            output_dir = output_dir if output_dir is not None else self.args.output_dir
            os.makedirs(output_dir, exist_ok=True)
            model.save_pretrained(output_dir)
            tok = getattr(self, "processing_class", None) or getattr(self, "tokenizer", None)
            if tok is not None:
                tok.save_pretrained(output_dir)
            return
        return orig(self, output_dir, state_dict)

    Trainer._save = _save
    Trainer._bigdl_save_patched = True


setup_training_env()
