"""One-call model + tokenizer loading used by the serving workers
(reference: python/llm/src/ipex_llm/transformers/loader.py:41-173 ``load_model`` / ``try_run_test_generation`` /
``get_model_answers`` warm-up)."""
from __future__ import annotations

import time

import torch

LLAMA_IDS = ["llama", "vicuna", "merged-baize"]


def load_model(model_path: str, device: str = "cpu", low_bit: str = "sym_int4", trust_remote_code: bool = True,
               speculative: bool = False, load_low_bit_model: bool = False):
    """-> (model, tokenizer).  ``device``: "cpu" or "cuda" (the reference's "xpu" is accepted as an alias)."""
    from transformers import AutoTokenizer
    from . import AutoModel, AutoModelForCausalLM
    device = "cuda" if device in ("xpu", "gpu") else device
    tokenizer_cls = AutoTokenizer
    model_kwargs = {"use_cache": True}
    if trust_remote_code:
        model_kwargs["trust_remote_code"] = True
    if speculative:
        low_bit = "bf16" if device == "cpu" else "fp16"
        model_kwargs["speculative"] = True
    tokenizer = tokenizer_cls.from_pretrained(model_path, trust_remote_code=trust_remote_code)
    t0 = time.time()
    auto = AutoModel if "chatglm" in model_path.lower() else AutoModelForCausalLM
    if load_low_bit_model:
        model = auto.load_low_bit(model_path, **{k: v for k, v in model_kwargs.items() if k != "use_cache"})
    elif low_bit == "bf16":
        model = auto.from_pretrained(model_path, load_in_low_bit="bf16", torch_dtype=torch.bfloat16, **model_kwargs)
    else:
        model = auto.from_pretrained(model_path, load_in_low_bit=low_bit, **model_kwargs)
    if device == "cuda":
        model = model.half().to("cuda") if low_bit != "bf16" else model.to("cuda")
    model = model.eval()
    print(f"Model loaded in {time.time() - t0:.1f}s")
    if device == "cuda":
        try_run_test_generation(model, tokenizer)
    return model, tokenizer


def try_run_test_generation(local_model, tokenizer, prompt: str = "Tell me about Intel Xeon."):
    """Warm-up generation: builds kernels' lazy state and the decode graph before the first request."""
    ids = tokenizer(prompt, return_tensors="pt").input_ids
    ids = ids.to(next(p for p in local_model.parameters() if p.numel() > 0).device)
    with torch.inference_mode():
        out = local_model.generate(ids, max_new_tokens=8, do_sample=False)
    return tokenizer.decode(out[0], skip_special_tokens=True)
