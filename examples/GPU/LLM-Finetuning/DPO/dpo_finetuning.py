"""DPO on an nf4 base model with LoRA adapters; the reference policy is the same model with the adapters off.
(reference: example/GPU/LLM-Finetuning/DPO/dpo_finetuning.py, which drives trl.DPOTrainer(ref_model=None))

    python dpo_finetuning.py --repo-id-or-model-path /models/Mistral-7B --dataset pairs.jsonl
pairs.jsonl: one {"prompt": ..., "chosen": ..., "rejected": ...} per line.
"""
import argparse
import json

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM
from ipex_llm.transformers.qlora import LoraConfig, get_peft_model, prepare_model_for_kbit_training
from bigdl_amd.transformers.dpo import DPOConfig, DPOTrainer, build_pair_batch

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--dataset", required=True)
    ap.add_argument("--output-path", default="./dpo-out")
    ap.add_argument("--micro-batch", type=int, default=4)
    ap.add_argument("--grad-accum", type=int, default=4)
    ap.add_argument("--max-steps", type=int, default=200)
    ap.add_argument("--beta", type=float, default=0.1)
    args = ap.parse_args()

    tok = AutoTokenizer.from_pretrained(args.repo_id_or_model_path)
    model = AutoModelForCausalLM.from_pretrained(args.repo_id_or_model_path, load_in_low_bit="nf4", optimize_model=False,
                                                 torch_dtype=torch.bfloat16, modules_to_not_convert=["lm_head"])
    model = prepare_model_for_kbit_training(model.to("cuda"), use_gradient_checkpointing=True)
    model = get_peft_model(model, LoraConfig(r=16, lora_alpha=16, lora_dropout=0.05, training_mode="qlora",
                                             target_modules=["k_proj", "gate_proj", "v_proj", "up_proj", "q_proj",
                                                             "o_proj", "down_proj"]))
    rows = [json.loads(l) for l in open(args.dataset)]
    trainer = DPOTrainer(model, DPOConfig(beta=args.beta, learning_rate=5e-5, gradient_accumulation_steps=args.grad_accum))
    for step in range(args.max_steps * args.grad_accum):
        chunk = [rows[(step * args.micro_batch + i) % len(rows)] for i in range(args.micro_batch)]
        batch = {k: v.cuda() for k, v in build_pair_batch(tok, chunk, max_length=1024, max_prompt_length=512).items()}
        m = trainer.step(batch)
        if step % (10 * args.grad_accum) == 0:
            print(step // args.grad_accum, {k: round(v, 4) for k, v in m.items()})
    model.save_pretrained(args.output_path)
