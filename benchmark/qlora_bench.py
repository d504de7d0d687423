"""QLoRA training-step benchmark: Llama-2-7B nf4 base (random-init, created directly in the device layout), LoRA r=8 on
q/k/v like the reference's example (python/llm/example/GPU/LLM-Finetuning/QLoRA/simple-example/qlora_finetuning.py:
77-110: per-device batch 4, bf16 compute, lm_head not converted), AdamW, synthetic token batches.

    python benchmark/qlora_bench.py --steps 10 --seq-len 256
    torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 8 benchmark/qlora_bench.py   # data parallel over RCCL

Prints one JSON line: s/it, samples/s (whole job), peak memory.  The reference's published figure for this setup is
1.40 s/it = 2.864 samples/s on one (unnamed) Intel GPU
(python/llm/example/GPU/LLM-Finetuning/QLoRA/simple-example/README.md:48-50)."""
import argparse
import json
import os
import time

import torch
from torch import nn


def random_lowbit_llama(cfg, qtype, device, skip=("lm_head",), seed=0):
    """HF LlamaForCausalLM whose linears are LowBitLinear with random quantized weights made on the device (no 27 GB
    fp32 detour through host quantization)."""
    from transformers import LlamaForCausalLM
    from bigdl_amd.models.decoder import _random_t16
    from bigdl_amd.transformers.low_bit_linear import LowBitLinear
    with torch.device("meta"):
        model = LlamaForCausalLM(cfg)
    gen = torch.Generator(device=device).manual_seed(seed)

    def fill(mod, prefix=""):
        for name, child in list(mod.named_children()):
            full = prefix + name
            if isinstance(child, nn.Linear) and name not in skip:
                new = LowBitLinear.__new__(LowBitLinear)
                nn.Module.__init__(new)
                n, k = child.out_features, child.in_features
                new.in_features, new.out_features = k, n
                new.in_len, new.out_len, new.weight_shape, new.weight_length = k, n, (n, k), n * k
                new.qtype, new.conver_to_half, new.mp_group, new.compute_dtype = qtype, True, None, torch.bfloat16
                new.optimize_lm_head, new.act_order = False, False
                new._parameters["weight"] = _random_t16(n, k, qtype, device, gen, scale=0.02)
                new._parameters["bias"] = None
                mod._modules[name] = new
            else:
                fill(child, full + ".")
    fill(model)
    for name, p in list(model.named_parameters()):
        if p.device.type == "meta":
            mod = model.get_submodule(name.rsplit(".", 1)[0])
            init = torch.ones if "norm" in name else (lambda *s, **k: torch.randn(*s, **k) * 0.02)
            mod._parameters[name.rsplit(".", 1)[1]] = nn.Parameter(
                init(p.shape, device=device, dtype=torch.bfloat16), requires_grad=False)
    for name, b in list(model.named_buffers()):
        if b.device.type == "meta":
            mod = model.get_submodule(name.rsplit(".", 1)[0]) if "." in name else model
            if name.endswith("inv_freq"):
                d = cfg.hidden_size // cfg.num_attention_heads
                mod._buffers[name.rsplit(".", 1)[-1]] = 1.0 / (10000.0 ** (torch.arange(0, d, 2, device=device).float() / d))
    return model


def _dense_cache_on():
    from bigdl_amd.transformers.low_bit_linear import train_dense_cache_enabled
    return train_dense_cache_enabled()


def main(process_group_ready: bool = False):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--seq-len", type=int, default=256)
    ap.add_argument("--low-bit", default="nf4")
    ap.add_argument("--model", default="llama2-7b", choices=["llama2-7b", "tiny"])
    ap.add_argument("--grad-ckpt", action="store_true")
    ap.add_argument("--full-targets", action="store_true",
                    help="LoRA on all seven projections (q, k, v, o, gate, up, down) as in the reference's alpaca-qlora "
                         "recipe (alpaca_qlora_finetuning.py:80-88) instead of q/k/v only")
    ap.add_argument("--graph", action="store_true",
                    help="forward + backward (+ the optimizer step on one GPU) replayed as one hipGraph "
                         "(bigdl_amd/transformers/graph_train.py)")
    args = ap.parse_args()
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 and not process_group_ready:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    from transformers import LlamaConfig
    from bigdl_amd import qtypes as Q
    from bigdl_amd.parallel.data_parallel import DataParallelAdapter
    from bigdl_amd.transformers.qlora import LoraConfig, get_peft_model, prepare_model_for_kbit_training
    if args.model == "llama2-7b":
        cfg = LlamaConfig(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                          num_attention_heads=32, num_key_value_heads=32, max_position_embeddings=4096)
    else:
        cfg = LlamaConfig(vocab_size=2048, hidden_size=512, intermediate_size=1024, num_hidden_layers=4,
                          num_attention_heads=8, num_key_value_heads=8, max_position_embeddings=1024)
    cfg._attn_implementation = "sdpa"
    model = random_lowbit_llama(cfg, Q.ggml_tensor_qtype[args.low_bit], dev)
    model.train()
    model = prepare_model_for_kbit_training(model, use_gradient_checkpointing=args.grad_ckpt)
    targets = ["q_proj", "k_proj", "v_proj"] + (["o_proj", "gate_proj", "up_proj", "down_proj"] if args.full_targets else [])
    model = get_peft_model(model, LoraConfig(r=8, lora_alpha=32, target_modules=targets,
                                             lora_dropout=0.05, bias="none", task_type="CAUSAL_LM", training_mode="qlora"))
    params = [p for p in model.parameters() if p.requires_grad]
    g = torch.Generator().manual_seed(rank)
    if args.graph:
        from bigdl_amd.parallel.data_parallel import broadcast_trainable
        from bigdl_amd.transformers.graph_train import GraphedTrainStep, capturable_adamw
        opt = capturable_adamw(params, lr=2e-5)
        broadcast_trainable(model, 0)
        graphed = GraphedTrainStep(model, opt, torch.randint(3, cfg.vocab_size, (args.batch, args.seq_len), generator=g).to(dev))

        def step():
            return graphed(torch.randint(3, cfg.vocab_size, (args.batch, args.seq_len), generator=g).to(dev))
    else:
        opt = torch.optim.AdamW(params, lr=2e-5)
        dp = DataParallelAdapter(model, opt)

        def step():
            ids = torch.randint(3, cfg.vocab_size, (args.batch, args.seq_len), generator=g).to(dev)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(input_ids=ids, labels=ids)
            dp.step(out.loss)
            return out.loss

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    dt = (time.perf_counter() - t0) / args.steps
    if rank == 0:
        print(json.dumps({"metric": "QLoRA finetune step time (s/it), %s %s, batch %d x %d tokens per GPU" %
                                    (args.low_bit, "Llama-2-7B" if args.model == "llama2-7b" else "test-size llama", args.batch, args.seq_len),
                          "value": round(dt, 4), "unit": "s/it", "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt * 1e3, 2), "higher_is_better": False, "scaling": "weak",
                          "vs_baseline": round(dt / 1.40, 4) if args.model == "llama2-7b" and not args.full_targets else None,
                          "dtype": "bf16 compute, %s base weights" % args.low_bit, "data": "synthetic",
                          "config": {"model": args.model + " (random-init)", "global_batch": world * args.batch,
                                     "seq_len": args.seq_len, "parallelism": f"dp{world}" if world > 1 else "single",
                                     "lora_targets": targets, "lora_r": 8, "hip_graph": bool(args.graph),
                                     "dense_weight_cache": _dense_cache_on()},
                          "bench": "qlora_train_step", "model": args.model, "low_bit": args.low_bit, "n_gpus": world,
                          "per_device_batch": args.batch, "seq_len": args.seq_len, "s_per_it": round(dt, 4),
                          "samples_per_s": round(world * args.batch / dt, 2),
                          "tokens_per_s": round(world * args.batch * args.seq_len / dt, 1),
                          "trainable_params": sum(p.numel() for p in params), "loss": round(float(loss.detach()), 4),
                          "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                          "reference_published": {"s_per_it": 1.40, "samples_per_s": 2.864,
                                                  "hardware": "1 Intel GPU (unnamed), english_quotes, batch 4"}}))
    if world > 1 and not process_group_ready:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
