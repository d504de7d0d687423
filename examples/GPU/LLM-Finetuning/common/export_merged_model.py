"""Merge a trained LoRA / QLoRA adapter into the full-precision base model and save a plain HuggingFace checkpoint
(reference: python/llm/example/GPU/LLM-Finetuning/QLoRA/export_merged_model.py + common/utils/util.py ``merge_adapter``).

    python export_merged_model.py --repo-id-or-model-path <base> --adapter_path <outputs/checkpoint-200> --output_path merged
"""
import argparse

import torch
from transformers import AutoModelForCausalLM, AutoTokenizer

from ipex_llm.transformers.qlora import PeftModel

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--adapter_path", required=True)
    ap.add_argument("--output_path", required=True)
    a = ap.parse_args()
    base = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, torch_dtype=torch.float16)     # un-quantized base
    model = PeftModel.from_pretrained(base, a.adapter_path)
    merged = model.merge_and_unload()                   # W += (alpha / r) * B @ A for every adapted linear
    merged.save_pretrained(a.output_path, safe_serialization=True)
    AutoTokenizer.from_pretrained(a.repo_id_or_model_path).save_pretrained(a.output_path)
    print(f"merged model written to {a.output_path}")
