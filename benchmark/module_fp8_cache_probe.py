"""Module-path decode step with the three KV caches a HuggingFace-style model can carry here: fp16 (DynamicNormalCache),
fp8 expanded to fp16 on every step (DynamicFp8Cache), fp8 read as bytes by the attention kernel (DynamicFp8RawCache).
Times ``--steps`` single-token forwards through the patched module graph (not the native engine) after a prompt of
``--in-len`` tokens; one JSON line per cache.  The reference's counterpart is the ``sdp_fp8`` call that attends to the
fp8 cache without restoring it (python/llm/src/ipex_llm/transformers/models/common.py:243-253).

    python benchmark/module_fp8_cache_probe.py --model llama2-7b --in-len 1024 --steps 64
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.dirname(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama2-7b")
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--in-len", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    a = ap.parse_args()
    from hf_api_at_size import SHAPES, random_model
    from bigdl_amd.optimize import optimize_model
    model = optimize_model(random_model(a.model, "cuda"), low_bit=a.low_bit).half().to("cuda")
    ids = torch.randint(3, SHAPES[a.model]["vocab_size"], (a.batch, a.in_len), device="cuda")
    for name, env in (("fp16", {"IPEX_LLM_QUANTIZE_KV_CACHE": "0"}),
                      ("fp8_expanded", {"IPEX_LLM_QUANTIZE_KV_CACHE": "1", "BIGDL_FP8_RAW_CACHE": "0"}),
                      ("fp8_bytes", {"IPEX_LLM_QUANTIZE_KV_CACHE": "1", "BIGDL_FP8_RAW_CACHE": "1"})):
        os.environ.update(env)
        with torch.no_grad():
            out = model(ids, use_cache=True)
            pkv = out.past_key_values
            tok = out.logits[:, -1:].argmax(-1)
            for i in range(a.steps + 8):
                if i == 8:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                tok = model(tok, past_key_values=pkv, use_cache=True).logits[:, -1:].argmax(-1)
            torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / a.steps
        print(json.dumps({"cache": name, "class": type(pkv).__name__, "model": a.model, "low_bit": a.low_bit, "batch": a.batch,
                          "ctx": a.in_len, "ms_per_step": round(ms, 3)}), flush=True)
        del pkv, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
