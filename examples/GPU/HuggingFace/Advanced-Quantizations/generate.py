"""Pre-quantized checkpoints: AutoGPTQ / AutoAWQ int4 models are repacked into asym_int4 at load time, GGUF files are
read directly (reference: python/llm/example/GPU/HuggingFace/Advanced-Quantizations/{AWQ,GPTQ,GGUF}/generate.py)."""
import argparse

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True, help="GPTQ / AWQ checkpoint directory or a .gguf file")
    ap.add_argument("--prompt", default="What is AI?")
    ap.add_argument("--n-predict", type=int, default=32)
    a = ap.parse_args()
    if a.repo_id_or_model_path.endswith(".gguf"):
        model, tok = AutoModelForCausalLM.from_gguf(a.repo_id_or_model_path)
    else:
        # quantization_config in the checkpoint's config.json selects the GPTQ / AWQ path; both land in asym_int4
        model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_4bit=True, torch_dtype=torch.float16,
                                                     trust_remote_code=True)
        tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path, trust_remote_code=True)
    model = model.half().to("cuda")
    ids = tok(a.prompt, return_tensors="pt").input_ids.to("cuda")
    with torch.inference_mode():
        out = model.generate(ids, max_new_tokens=a.n_predict)
    print(tok.decode(out[0], skip_special_tokens=True))
