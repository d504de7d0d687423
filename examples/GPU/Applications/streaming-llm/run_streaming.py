"""Endless multi-turn chat on a bounded KV cache with attention sinks (StreamingLLM), on the native engine.

The reference runs this application through the external ``streaming_llm`` package on a low-bit model
(python/llm/example/GPU/Applications/streaming-llm/run_streaming_llama.py: ``enable_streaming_llm(model, start_size,
recent_size)`` + ``kv_cache.evict_for_space``).  Here the engine does it itself: ``NativeDecoder.evict_middle`` keeps the
first ``--start-size`` keys and the last ``--recent-size``, rotates the kept keys back to their positions inside the
cache, and generation carries on from there; every turn is appended to the same cache.

    python run_streaming.py --repo-id-or-model-path /models/Llama-2-7b-chat-hf --prompts prompts.jsonl \
        [--start-size 4 --recent-size 2000 --max-gen-len 512]
``prompts.jsonl``: one JSON object per line with a ``turns`` list of user messages (the MT-Bench file the reference
downloads has that shape).
"""
import argparse
import json

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM


@torch.no_grad()
def stream_turn(eng, tok, ids, max_gen_len, eos):
    logits = eng.prefill(ids)                                  # appended behind what the cache already holds
    nxt = logits[:, -1].argmax(-1).view(1, 1)
    out = []
    for _ in range(max_gen_len):
        t = int(nxt)
        if t in eos:
            break
        out.append(t)
        print(tok.decode(out[-1:], skip_special_tokens=True), end="", flush=True)
        _, n2 = eng.decode(nxt)
        nxt = n2.view(1, 1)
    print()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompts", required=True)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--start-size", type=int, default=4)
    ap.add_argument("--recent-size", type=int, default=2000)
    ap.add_argument("--max-gen-len", type=int, default=512)
    a = ap.parse_args()
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit, trust_remote_code=True)
    model = model.half().to("cuda")
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path, trust_remote_code=True)
    eng = model.native_engine()                                # the engine behind model.generate
    capacity = a.start_size + a.recent_size + 2 * a.max_gen_len + 1024
    eng.ensure_cache(1, capacity, fp8=False)                   # eviction re-rotates keys: fp16 cache
    eng.cache.reset()
    eos = {tok.eos_token_id}
    with open(a.prompts) as f:
        turns = [t for line in f if line.strip() for t in json.loads(line)["turns"]]
    for i, user in enumerate(turns):
        text = f"USER: {user}\n\nASSISTANT: "
        print(f"\n{text}", end="")
        ids = tok(text, return_tensors="pt", add_special_tokens=(i == 0)).input_ids.cuda()
        need = ids.shape[1] + a.max_gen_len
        if eng.cache.host_len + need > capacity:
            room = capacity - need - a.start_size
            dropped = eng.evict_middle(a.start_size, min(a.recent_size, room))
            print(f"[evicted {dropped} positions, cache now {eng.cache.host_len}]")
        stream_turn(eng, tok, ids, a.max_gen_len, eos)


if __name__ == "__main__":
    main()
