#!/bin/bash
# ReLoRA fine-tuning of Llama-2-7B on Alpaca-format data, data parallel over the 8 GPUs of a node (base weights: nf4).
# Same script as QLoRA with --training_mode relora (reference: python/llm/example/GPU/LLM-Finetuning/ReLoRA/).
torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 8 \
  ../QLoRA/alpaca_qlora_finetuning.py --base_model "${1:?model path}" --data_path "${2:?alpaca json}" \
  --training_mode relora --output_dir ./relora-out --micro_batch_size 4 --batch_size 128
