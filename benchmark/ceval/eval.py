"""C-Eval (Chinese multiple-choice exam suite) accuracy of a low-bit model.

Reference: python/llm/dev/benchmark/ceval/eval.py:225-338 + evaluators/{llama,chatglm,qwen}.py: per subject, a few-shot
prompt built from the ``dev`` split, the answer is the option letter with the highest next-token logit, results are
aggregated per category into a JSON file.  Data layout (no network: point ``--eval-data-path`` at an unpacked copy):
``<data>/val/<subject>_val.csv`` and ``<data>/dev/<subject>_dev.csv`` with columns id,question,A,B,C,D,answer[,explanation].
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
from typing import Dict, List

import torch

CHOICES = ["A", "B", "C", "D"]


def format_example(row: dict, with_answer: bool) -> str:
    s = row["question"] + "".join(f"\n{c}. {row[c]}" for c in CHOICES) + "\n答案："
    return s + (row["answer"] + "\n\n" if with_answer else "")


def build_prompt(subject: str, dev_rows: List[dict], row: dict, k: int) -> str:
    head = f"以下是中国关于{subject}考试的单项选择题，请选出其中的正确答案。\n\n"
    return head + "".join(format_example(r, True) for r in dev_rows[:k]) + format_example(row, False)


def option_token_ids(tokenizer) -> List[int]:
    ids = []
    for c in CHOICES:
        t = tokenizer(c, add_special_tokens=False).input_ids
        ids.append(t[-1])
    return ids


@torch.no_grad()
def predict(model, tokenizer, prompt: str, opt_ids: List[int], device, max_len: int = 2048) -> str:
    ids = tokenizer(prompt, return_tensors="pt").input_ids[:, -max_len:].to(device)
    out = model(ids)
    logits = (out.logits if hasattr(out, "logits") else out[0])[0, -1].float()
    return CHOICES[int(logits[opt_ids].argmax())]


def read_csv(path: str) -> List[dict]:
    with open(path, newline="", encoding="utf-8") as f:
        return list(csv.DictReader(f))


def evaluate(model, tokenizer, data_path: str, split: str = "val", k: int = 5, device="cpu", limit: int = 0) -> Dict[str, dict]:
    opt = option_token_ids(tokenizer)
    res = {}
    for f in sorted(glob.glob(os.path.join(data_path, split, f"*_{split}.csv"))):
        subject = os.path.basename(f)[: -len(f"_{split}.csv")]
        dev_f = os.path.join(data_path, "dev", f"{subject}_dev.csv")
        dev = read_csv(dev_f) if os.path.exists(dev_f) else []
        rows = read_csv(f)
        rows = rows[:limit] if limit else rows
        preds = [predict(model, tokenizer, build_prompt(subject, dev, r, k), opt, device) for r in rows]
        if rows and "answer" in rows[0] and rows[0]["answer"]:
            correct = sum(int(p == r["answer"]) for p, r in zip(preds, rows))
            res[subject] = {"n": len(rows), "acc": correct / max(1, len(rows))}
        else:                                  # test split: answers are withheld, keep the predictions for submission
            res[subject] = {"n": len(rows), "predictions": {r["id"]: p for r, p in zip(rows, preds)}}
    scored = [v for v in res.values() if "acc" in v]
    if scored:
        res["_average"] = {"n": sum(v["n"] for v in scored),
                           "acc": sum(v["acc"] * v["n"] for v in scored) / max(1, sum(v["n"] for v in scored))}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_path", required=True)
    ap.add_argument("--eval_type", default="validation", choices=["validation", "test"])
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--eval_data_path", default="data")
    ap.add_argument("--qtype", default="sym_int4")
    ap.add_argument("--shots", type=int, default=5)
    ap.add_argument("--limit", type=int, default=0)
    a = ap.parse_args()
    from transformers import AutoTokenizer
    from ipex_llm.transformers import AutoModelForCausalLM
    model = AutoModelForCausalLM.from_pretrained(a.model_path, load_in_low_bit=a.qtype, trust_remote_code=True).eval()
    if a.device != "cpu":
        model = model.half().to(a.device)
    tok = AutoTokenizer.from_pretrained(a.model_path, trust_remote_code=True)
    res = evaluate(model, tok, a.eval_data_path, "val" if a.eval_type == "validation" else "test", a.shots, a.device, a.limit)
    out = f"ceval_{os.path.basename(a.model_path.rstrip('/'))}_{a.qtype}.json"
    json.dump(res, open(out, "w"), ensure_ascii=False, indent=1)
    print(json.dumps(res.get("_average", {})), "->", out)


if __name__ == "__main__":
    main()
