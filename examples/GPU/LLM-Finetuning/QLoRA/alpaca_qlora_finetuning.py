"""QLoRA fine-tuning on Alpaca-format data, data parallel over the GPUs of a node:
    torchrun --standalone --nproc-per-node 8 alpaca_qlora_finetuning.py --base_model <path> --data_path alpaca.json
(reference: python/llm/example/GPU/LLM-Finetuning/QLoRA/alpaca-qlora/alpaca_qlora_finetuning.py)."""
import argparse
import json
import os

import torch
from transformers import AutoTokenizer

from bigdl_amd.parallel import init_distributed
from bigdl_amd.parallel.data_parallel import DataParallelAdapter
from ipex_llm.transformers import AutoModelForCausalLM
from ipex_llm.transformers.qlora import LoraConfig, get_peft_model, prepare_model_for_kbit_training

PROMPT = ("Below is an instruction that describes a task{inp}. Write a response that appropriately completes the request."
          "\n\n### Instruction:\n{instruction}\n\n{input_block}### Response:\n")


def build_examples(path, tok, cutoff_len):
    rows = json.load(open(path))
    out = []
    for r in rows:
        inp = r.get("input", "")
        prompt = PROMPT.format(inp=", paired with an input that provides further context" if inp else "",
                               instruction=r["instruction"], input_block=f"### Input:\n{inp}\n\n" if inp else "")
        p_ids = tok(prompt, add_special_tokens=True).input_ids
        a_ids = tok(r["output"], add_special_tokens=False).input_ids + [tok.eos_token_id]
        ids = (p_ids + a_ids)[:cutoff_len]
        labels = ([-100] * len(p_ids) + a_ids)[:cutoff_len]
        out.append((ids, labels))
    return out


def collate(batch, pad, length=None):
    n = length or max(len(i) for i, _ in batch)
    ids = torch.full((len(batch), n), pad)
    labels = torch.full((len(batch), n), -100)
    mask = torch.zeros(len(batch), n, dtype=torch.long)
    for b, (i, l) in enumerate(batch):
        ids[b, :len(i)] = torch.tensor(i)
        labels[b, :len(l)] = torch.tensor(l)
        mask[b, :len(i)] = 1
    return ids, labels, mask


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--base_model", required=True)
    ap.add_argument("--data_path", required=True)
    ap.add_argument("--output_dir", default="./qlora-out")
    ap.add_argument("--training_mode", default="qlora", choices=["qlora", "qalora", "lora", "relora"])
    ap.add_argument("--micro_batch_size", type=int, default=2)
    ap.add_argument("--batch_size", type=int, default=128)
    ap.add_argument("--num_epochs", type=int, default=1)
    ap.add_argument("--learning_rate", type=float, default=3e-5)
    ap.add_argument("--cutoff_len", type=int, default=256)
    ap.add_argument("--lora_r", type=int, default=8)
    ap.add_argument("--lora_alpha", type=int, default=16)
    ap.add_argument("--lora_dropout", type=float, default=0.05)
    ap.add_argument("--lora_target_modules", nargs="+",
                    default=["q_proj", "k_proj", "v_proj", "o_proj", "up_proj", "down_proj", "gate_proj"])
    ap.add_argument("--relora_steps", type=int, default=300)
    ap.add_argument("--hip_graph", action="store_true",
                    help="pad every batch to cutoff_len and replay forward + backward as one hipGraph "
                         "(bigdl_amd/transformers/graph_train.py); no activation checkpointing in this mode")
    a = ap.parse_args()
    rank, world, device = init_distributed()
    low_bit = {"qlora": "nf4", "qalora": "sym_int4", "lora": "bf16", "relora": "nf4"}[a.training_mode]
    model = AutoModelForCausalLM.from_pretrained(a.base_model, load_in_low_bit=low_bit, optimize_model=False,
                                                 torch_dtype=torch.bfloat16, modules_to_not_convert=["lm_head"])
    model = model.to(device)
    model = prepare_model_for_kbit_training(model, use_gradient_checkpointing=not a.hip_graph)
    cfg = LoraConfig(r=a.lora_r, lora_alpha=a.lora_alpha, target_modules=a.lora_target_modules,
                     lora_dropout=a.lora_dropout, bias="none", task_type="CAUSAL_LM", training_mode=a.training_mode)
    model = get_peft_model(model, cfg)
    if rank == 0:
        model.print_trainable_parameters()
    tok = AutoTokenizer.from_pretrained(a.base_model)
    tok.pad_token_id = tok.pad_token_id if tok.pad_token_id is not None else 0
    data = build_examples(a.data_path, tok, a.cutoff_len)[rank::world]
    accum = max(1, a.batch_size // (a.micro_batch_size * world))          # global batch is divided over the ranks
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=a.learning_rate)
    dp = DataParallelAdapter(model, opt, grad_accum=accum, max_grad_norm=1.0)
    model.train()
    step = 0
    if a.hip_graph:
        from bigdl_amd.transformers.graph_train import GraphedTrainStep
        first = tuple(t.to(device) for t in collate(data[:a.micro_batch_size], tok.pad_token_id, a.cutoff_len))
        graphed = GraphedTrainStep(model, opt, first[0], example_labels=first[1], example_mask=first[2], grad_accum=accum,
                                   max_grad_norm=1.0)
        for epoch in range(a.num_epochs):
            for i in range(0, len(data) - a.micro_batch_size + 1, a.micro_batch_size):
                ids, labels, mask = (t.to(device) for t in collate(data[i:i + a.micro_batch_size], tok.pad_token_id, a.cutoff_len))
                loss = graphed(ids, labels, mask)
                if graphed.stepped:
                    step += 1
                    if rank == 0 and step % 10 == 0:
                        print(f"epoch {epoch} step {step} loss {loss.item():.4f}", flush=True)
        a.num_epochs = 0
    for epoch in range(a.num_epochs):
        for i in range(0, len(data) - a.micro_batch_size + 1, a.micro_batch_size):
            ids, labels, mask = (t.to(device) for t in collate(data[i:i + a.micro_batch_size], tok.pad_token_id))
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = model(input_ids=ids, attention_mask=mask, labels=labels).loss
            if dp.step(loss):
                step += 1
                if a.training_mode == "relora" and step % a.relora_steps == 0:
                    from ipex_llm.transformers.relora import merge_and_reinit, reset_optimizer
                    reset_optimizer(opt, merge_and_reinit(model))
                if rank == 0 and step % 10 == 0:
                    print(f"epoch {epoch} step {step} loss {loss.item():.4f}", flush=True)
    if rank == 0:
        os.makedirs(a.output_dir, exist_ok=True)
        model.save_pretrained(a.output_dir)
