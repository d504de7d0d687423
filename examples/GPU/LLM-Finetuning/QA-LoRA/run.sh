#!/bin/bash
# QA-LoRA fine-tuning of Llama-2-7B on Alpaca-format data, data parallel over the 8 GPUs of a node (base weights: sym_int4).
# Same script as QLoRA with --training_mode qalora (reference: python/llm/example/GPU/LLM-Finetuning/QA-LoRA/).
torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 8 \
  ../QLoRA/alpaca_qlora_finetuning.py --base_model "${1:?model path}" --data_path "${2:?alpaca json}" \
  --training_mode qalora --output_dir ./qalora-out --micro_batch_size 4 --batch_size 128
