#!/bin/bash
# LoRA fine-tuning of Llama-2-7B on Alpaca-format data, data parallel over the 8 GPUs of a node (base weights: bf16).
# Same script as QLoRA with --training_mode lora (reference: python/llm/example/GPU/LLM-Finetuning/LoRA/).
torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 8 \
  ../QLoRA/alpaca_qlora_finetuning.py --base_model "${1:?model path}" --data_path "${2:?alpaca json}" \
  --training_mode lora --output_dir ./lora-out --micro_batch_size 4 --batch_size 128
