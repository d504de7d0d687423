"""Encoder-decoder generation (flan-t5, mt5, ...) in low-bit on one MI355X (reference:
python/llm/example/GPU/HuggingFace/LLM/flan-t5/generate.py).  The reference has to leave the feed-forward output
projection ``wo`` and the head unconverted; here both can be low-bit (the converter rebinds the feed-forward block so
that it does not cast activations to the packed weight's byte type, and a seq2seq head scores every position)."""
import argparse
import time

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForSeq2SeqLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt", default="Translate to German: My name is Arthur")
    ap.add_argument("--n-predict", type=int, default=32)
    ap.add_argument("--low-bit", default="sym_int4")
    a = ap.parse_args()
    model = AutoModelForSeq2SeqLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit, trust_remote_code=True)
    model = model.half().to("cuda")
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path, trust_remote_code=True)
    ids = tok(a.prompt, return_tensors="pt").input_ids.to("cuda")
    with torch.inference_mode():
        model.generate(ids, max_new_tokens=a.n_predict, use_cache=True)          # warm-up
        torch.cuda.synchronize()
        t0 = time.time()
        out = model.generate(ids, max_new_tokens=a.n_predict, use_cache=True)
        torch.cuda.synchronize()
    print(f"Inference time: {time.time() - t0:.3f} s")
    print(tok.decode(out[0], skip_special_tokens=True))
