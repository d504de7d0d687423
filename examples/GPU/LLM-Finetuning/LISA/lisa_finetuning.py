"""LISA: full-parameter fine-tuning with only a few randomly chosen decoder layers unfrozen at a time
(reference: python/llm/example/GPU/LLM-Finetuning/LISA/lisa_finetuning.py, callback ipex_llm/transformers/lisa.py)."""
import argparse

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM
from ipex_llm.transformers.lisa import DynamicLayerActivationCallback

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--base_model", required=True)
    ap.add_argument("--text_file", required=True, help="plain text, one training sample per line")
    ap.add_argument("--lisa_activated_layers", type=int, default=1)
    ap.add_argument("--lisa_interval_steps", type=int, default=20)
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    model = AutoModelForCausalLM.from_pretrained(a.base_model, load_in_low_bit="bf16", optimize_model=False,
                                                 torch_dtype=torch.bfloat16).to("cuda").train()
    tok = AutoTokenizer.from_pretrained(a.base_model)
    lisa = DynamicLayerActivationCallback(n_layers=a.lisa_activated_layers, interval_steps=a.lisa_interval_steps, model=model)
    lines = [l.strip() for l in open(a.text_file) if l.strip()]
    opt = None
    for step in range(a.steps):
        if step % a.lisa_interval_steps == 0:
            lisa.switch_active_layers()                       # re-draw the unfrozen layers, rebuild the optimizer
            opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=5e-6)
        ids = tok(lines[step % len(lines)], return_tensors="pt", truncation=True, max_length=512).input_ids.to("cuda")
        loss = model(input_ids=ids, labels=ids).loss
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        if step % 10 == 0:
            print(f"step {step}: loss {loss.item():.4f}")
