"""Mean generation speed over an answer file written by eagle_generate.py (or by a plain-generation run in the same
layout): tokens per second per question, averaged.  Counterpart of the reference's
python/llm/example/GPU/Speculative-Decoding/EAGLE/evaluation/speed.py."""
import argparse
import json


def summarize(path, tokenizer=None):
    speeds, total_time, total_tokens = [], 0.0, 0
    for line in open(path, encoding="utf-8"):
        if not line.strip():
            continue
        choice = json.loads(line)["choices"][0]
        if "new_tokens" in choice:
            tokens = sum(choice["new_tokens"])
        else:                                  # answer files without token counts: count with the tokenizer
            tokens = sum(len(tokenizer(t).input_ids) - 1 for t in choice["turns"])
        seconds = sum(choice["wall_time"])
        speeds.append(tokens / seconds)
        total_time += seconds
        total_tokens += tokens
    return {"questions": len(speeds), "total_time_s": total_time, "total_tokens": total_tokens,
            "mean_tokens_per_s": sum(speeds) / max(1, len(speeds))}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--jsonl-file", required=True)
    ap.add_argument("--base-model-path", default=None, help="tokenizer for answer files that carry no token counts")
    a = ap.parse_args()
    tok = None
    if a.base_model_path:
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(a.base_model_path)
    print(json.dumps(summarize(a.jsonl_file, tok)))
