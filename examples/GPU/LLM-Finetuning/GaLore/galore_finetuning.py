"""Full-parameter fine-tuning with GaLore (rank-r projected Adam state) in bf16.
(reference: example/GPU/LLM-Finetuning/GaLore/galore_finetuning.py)

    python galore_finetuning.py --repo-id-or-model-path /models/Llama-2-7b-hf --data-path alpaca.json
"""
import argparse
import json

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM
from bigdl_amd.transformers.galore import GaLoreAdamW, galore_param_groups

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--data-path", required=True)
    ap.add_argument("--output-dir", default="./galore-out")
    ap.add_argument("--rank", type=int, default=1024)
    ap.add_argument("--max-steps", type=int, default=100)
    ap.add_argument("--micro-batch", type=int, default=2)
    ap.add_argument("--cutoff-len", type=int, default=1024)
    args = ap.parse_args()
    tok = AutoTokenizer.from_pretrained(args.repo_id_or_model_path)
    tok.pad_token = tok.pad_token or tok.eos_token
    # bf16 weights, optimized kernels off for training (dense matmuls carry the gradients)
    model = AutoModelForCausalLM.from_pretrained(args.repo_id_or_model_path, load_in_low_bit="bf16", optimize_model=False,
                                                 torch_dtype=torch.bfloat16).to("cuda")
    model.gradient_checkpointing_enable()
    model.train()
    opt = GaLoreAdamW(galore_param_groups(model, ["attn", "mlp"], rank=args.rank, update_proj_gap=200, scale=2.0),
                      lr=1e-5, weight_decay=0.0)
    rows = json.load(open(args.data_path))
    for step in range(args.max_steps):
        chunk = rows[step * args.micro_batch:(step + 1) * args.micro_batch] or rows[: args.micro_batch]
        text = [f"{r['instruction']}\n{r.get('input', '')}\n{r['output']}" for r in chunk]
        enc = tok(text, return_tensors="pt", padding=True, truncation=True, max_length=args.cutoff_len).to("cuda")
        labels = enc.input_ids.masked_fill(enc.attention_mask == 0, -100)
        loss = model(**enc, labels=labels).loss
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        if step % 10 == 0:
            print(step, round(loss.item(), 4), f"optimizer state {opt.state_numel() * 2 / 2**30:.2f} GiB (bf16)")
    model.save_pretrained(args.output_dir)
