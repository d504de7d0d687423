"""An unmodified HuggingFace + PEFT training script runs on low-bit weights after one line:
``llm_patch(train=True)`` rebinds transformers' AutoModel classes, peft's ``get_peft_model`` /
``prepare_model_for_kbit_training`` / ``LoraConfig`` and ``TrainingArguments`` to this package's versions
(reference: python/llm/example/GPU/LLM-Finetuning/HF-PEFT/, ipex_llm/llm_patching.py:35-85)."""
from ipex_llm import llm_patch

llm_patch(train=True)

# --- from here on: stock HF / PEFT code ------------------------------------------------------------------
import sys  # noqa: E402

import torch  # noqa: E402
import transformers  # noqa: E402
from peft import LoraConfig, get_peft_model, prepare_model_for_kbit_training  # noqa: E402
from transformers import AutoModelForCausalLM, AutoTokenizer, BitsAndBytesConfig  # noqa: E402

if __name__ == "__main__":
    path, text_file = sys.argv[1], sys.argv[2]
    bnb = BitsAndBytesConfig(load_in_4bit=True, bnb_4bit_quant_type="nf4", bnb_4bit_use_double_quant=False,
                             bnb_4bit_compute_dtype=torch.bfloat16)
    model = AutoModelForCausalLM.from_pretrained(path, quantization_config=bnb).to("cuda")
    model = prepare_model_for_kbit_training(model)
    model = get_peft_model(model, LoraConfig(r=8, lora_alpha=32, target_modules=["q_proj", "k_proj", "v_proj"],
                                             lora_dropout=0.05, bias="none", task_type="CAUSAL_LM"))
    tok = AutoTokenizer.from_pretrained(path)
    tok.pad_token = tok.pad_token or tok.eos_token
    data = [tok(l.strip(), truncation=True, max_length=256) for l in open(text_file) if l.strip()]
    trainer = transformers.Trainer(
        model=model, train_dataset=data,
        args=transformers.TrainingArguments(per_device_train_batch_size=4, gradient_accumulation_steps=1, warmup_steps=20,
                                            max_steps=200, learning_rate=2e-5, bf16=True, logging_steps=20,
                                            output_dir="outputs", optim="adamw_torch"),
        data_collator=transformers.DataCollatorForLanguageModeling(tok, mlm=False))
    trainer.train()
