"""QLoRA with ZeRO-partitioned optimizer state (stage 2) or parameters (stage 3) over the GPUs of one node.
(reference: example/GPU/LLM-Finetuning/QLoRA/alpaca-qlora/deepspeed_zero2.json, deepspeed_zero3.json)

    torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 8 qlora_zero.py --base /models/Llama-2-7b-hf --stage 2
"""
import argparse
import os

import torch
import torch.distributed as dist

from ipex_llm.transformers import AutoModelForCausalLM
from ipex_llm.transformers.qlora import LoraConfig, get_peft_model, prepare_model_for_kbit_training
from bigdl_amd.parallel.zero import ZeroOptimizer

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", required=True)
    ap.add_argument("--stage", type=int, default=2, choices=(1, 2, 3))
    ap.add_argument("--offload", action="store_true", help="optimizer state + fp32 master shard in pinned host memory")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--micro-batch", type=int, default=4)
    ap.add_argument("--seq-len", type=int, default=256)
    args = ap.parse_args()
    rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl")
    model = AutoModelForCausalLM.from_pretrained(args.base, load_in_low_bit="nf4", optimize_model=False,
                                                 torch_dtype=torch.bfloat16, modules_to_not_convert=["lm_head"])
    model = prepare_model_for_kbit_training(model.to("cuda"), use_gradient_checkpointing=True)
    model = get_peft_model(model, LoraConfig(r=8, lora_alpha=16, lora_dropout=0.05, training_mode="qlora",
                                             target_modules=["q_proj", "k_proj", "v_proj", "o_proj", "up_proj",
                                                             "down_proj", "gate_proj"]))
    opt = ZeroOptimizer(model.parameters(), lambda ps: torch.optim.AdamW(ps, lr=2e-4), max_grad_norm=0.3,
                        offload=args.offload, shard_params=args.stage == 3).attach(model)
    vocab = model.config.vocab_size
    g = torch.Generator().manual_seed(dist.get_rank())
    for step in range(args.steps):
        ids = torch.randint(0, vocab, (args.micro_batch, args.seq_len), generator=g).cuda()   # synthetic batch
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = model(input_ids=ids, labels=ids).loss
        loss.backward()
        opt.step()
        if dist.get_rank() == 0:
            print(f"step {step} loss {loss.item():.4f}")
    torch.save(opt.state_dict(), f"zero_rank{dist.get_rank()}.pt")
    dist.destroy_process_group()
