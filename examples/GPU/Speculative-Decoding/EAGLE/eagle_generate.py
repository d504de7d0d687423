"""EAGLE-2 speculative sampling with a low-bit base model and a converted draft layer.

Needs the EAGLE package (https://github.com/SafeAILab/EAGLE, ``pip install -e .``) and its draft-head checkpoint for
the base model; neither ships with this repository.  What this script adds is what the reference's example adds
(python/llm/example/GPU/Speculative-Decoding/EAGLE/evaluation/gen_ea_answer_llama2chat_e2_ipex_optimize.py:309-329):
the base model inside ``EaModel`` is replaced by its 4-bit conversion and the draft layer goes through
``optimize_model``; generation itself is EAGLE's own ``eagenerate`` (tree drafts, one verification pass per round).

    python eagle_generate.py --base-model-path meta-llama/Llama-2-7b-chat-hf --ea-model-path yuhuili/EAGLE-llama2-chat-7B \
        --question-file mt_bench_questions.jsonl --answer-file answers.jsonl
    python speed.py --base-model-path meta-llama/Llama-2-7b-chat-hf --jsonl-file answers.jsonl

Each line of the question file is ``{"question_id": ..., "turns": ["...", ...]}`` (MT-bench layout)."""
import argparse
import json
import time

import torch


def load(base_path, ea_path, low_bit, total_token, depth, top_k):
    from eagle.model.ea_model import EaModel                      # external package
    from ipex_llm import optimize_model
    from ipex_llm.transformers import AutoModelForCausalLM
    model = EaModel.from_pretrained(base_model_path=base_path, ea_model_path=ea_path, total_token=total_token, depth=depth,
                                    top_k=top_k, torch_dtype=torch.float16, low_cpu_mem_usage=True)
    if low_bit != "fp16":
        # the verification passes run on the low-bit base model ...
        model.base_model = AutoModelForCausalLM.from_pretrained(base_path, load_in_low_bit=low_bit, use_cache=True,
                                                                torch_dtype=torch.float16, optimize_model=True)
        # ... and the one-layer draft head is converted as well
        model.ea_layer = optimize_model(model.ea_layer, low_bit=low_bit)
    model = model.to("cuda")
    for name in ("tree_mask_init", "position_ids"):                # buffers EAGLE keeps as plain attributes
        t = getattr(model.ea_layer, name, None)
        if torch.is_tensor(t):
            setattr(model.ea_layer, name, t.to("cuda"))
    return model.eval()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base-model-path", required=True)
    ap.add_argument("--ea-model-path", required=True)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--question-file", required=True)
    ap.add_argument("--answer-file", default="answers.jsonl")
    ap.add_argument("--max-new-token", type=int, default=512)
    ap.add_argument("--total-token", type=int, default=60, help="draft tree size")
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--temperature", type=float, default=0.0)
    a = ap.parse_args()
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(a.base_model_path)
    model = load(a.base_model_path, a.ea_model_path, a.low_bit, a.total_token, a.depth, a.top_k)
    questions = [json.loads(line) for line in open(a.question_file) if line.strip()]
    with open(a.answer_file, "w") as out:
        for qi, q in enumerate(questions):
            history, turns, walls, new_tokens = "", [], [], []
            for turn in q["turns"]:
                prompt = f"{history}[INST] {turn} [/INST]"
                ids = tok([prompt], return_tensors="pt").input_ids.to("cuda")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with torch.inference_mode():
                    res = model.eagenerate(ids, temperature=a.temperature, max_new_tokens=a.max_new_token, log=True)
                torch.cuda.synchronize()
                wall = time.perf_counter() - t0
                out_ids = res[0] if isinstance(res, tuple) else res
                text = tok.decode(out_ids[0, ids.shape[1]:], skip_special_tokens=True).strip()
                turns.append(text)
                walls.append(wall)
                new_tokens.append(int(out_ids.shape[1] - ids.shape[1]))
                history = f"{prompt} {text} </s><s>"
            if qi == 0:
                continue                                              # warm-up question: kernels and caches settle
            out.write(json.dumps({"question_id": q.get("question_id", qi),
                                  "choices": [{"turns": turns, "wall_time": walls, "new_tokens": new_tokens}]}) + "\n")
            print(f"question {q.get('question_id', qi)}: {sum(new_tokens) / sum(walls):.1f} tokens/s", flush=True)


if __name__ == "__main__":
    main()
