"""LongBench with and without SnapKV-style KV compression.

Reference: python/llm/dev/benchmark/LongBench/{pred.py,eval.py,metrics.py,config.yaml} and the ablation configs
``config/ablation_c{512,1024,2048,4096}_w32_k7_maxpool.json`` (window 32, pooling kernel 7, capacity 512-4096).
Without network access the tasks come from local JSONL files (LongBench's own format: ``context``, ``input``,
``answers``, ``all_classes``, ``length``); prompts longer than the model limit are truncated in the middle, as upstream
does.  Metrics implemented here: token-level F1 (qa tasks), ROUGE-L F-measure (summarisation), classification and
exact-count scores.

    python benchmark/longbench/pred.py --model <path> --data-dir <dir with <dataset>.jsonl> --datasets qasper trec \
        --compress c512 c1024 --full-kv
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import re
import string
from typing import Dict, List

PROMPTS = {
    "qasper": "You are given a scientific article and a question. Answer the question as concisely as you can, using a single phrase or sentence if possible. If the question cannot be answered based on the information in the article, write \"unanswerable\".\n\nArticle: {context}\n\nQuestion: {input}\n\nAnswer:",
    "hotpotqa": "Answer the question based on the given passages. Only give me the answer and do not output any other words.\n\nThe following are given passages.\n{context}\n\nQuestion: {input}\nAnswer:",
    "multi_news": "You are given several news passages. Write a one-page summary of all news.\n\nNews:\n{context}\n\nSummary:",
    "trec": "Please determine the type of the question below. Here are some examples of questions.\n\n{context}\n{input}",
    "passage_count": "There are some paragraphs below sourced from Wikipedia. Some of them may be duplicates. Please carefully read these paragraphs and determine how many unique paragraphs there are after removing duplicates.\n\n{context}\n\nThe final answer is: ",
    "lcc": "Please complete the code given below.\n{context}Next line of code:\n",
}
MAX_GEN = {"qasper": 128, "hotpotqa": 32, "multi_news": 512, "trec": 64, "passage_count": 32, "lcc": 64}
METRIC = {"qasper": "f1", "hotpotqa": "f1", "multi_news": "rouge", "trec": "classification", "passage_count": "count",
          "lcc": "code_sim"}
COMPRESS = {"c512": 512, "c1024": 1024, "c2048": 2048, "c4096": 4096}       # window 32, kernel 7, maxpool


# ------------------------------------------------------------------ metrics
def normalize_answer(s: str) -> str:
    s = "".join(ch for ch in s.lower() if ch not in set(string.punctuation))
    s = re.sub(r"\b(a|an|the)\b", " ", s)
    return " ".join(s.split())


def f1_score(pred: str, gold: str) -> float:
    p, g = normalize_answer(pred).split(), normalize_answer(gold).split()
    common = collections.Counter(p) & collections.Counter(g)
    same = sum(common.values())
    if same == 0:
        return 0.0
    prec, rec = same / len(p), same / len(g)
    return 2 * prec * rec / (prec + rec)


def _lcs(a: List[str], b: List[str]) -> int:
    prev = [0] * (len(b) + 1)
    for x in a:
        cur = [0]
        for j, y in enumerate(b, 1):
            cur.append(prev[j - 1] + 1 if x == y else max(prev[j], cur[j - 1]))
        prev = cur
    return prev[-1]


def rouge_l(pred: str, gold: str) -> float:
    p, g = pred.split(), gold.split()
    if not p or not g:
        return 0.0
    l = _lcs(p, g)
    if l == 0:
        return 0.0
    prec, rec = l / len(p), l / len(g)
    return 2 * prec * rec / (prec + rec)


def classification_score(pred: str, gold: str, all_classes: List[str]) -> float:
    hits = [c for c in all_classes if c in pred]
    hits = [c for c in hits if not (c in gold and c != gold)]
    return 1.0 / len(hits) if gold in hits else 0.0


def count_score(pred: str, gold: str) -> float:
    nums = re.findall(r"\d+", pred)
    return sum(n == str(gold) for n in nums) / len(nums) if nums else 0.0


def code_sim_score(pred: str, gold: str) -> float:
    import difflib
    line = next((l for l in pred.lstrip("\n").split("\n") if l.strip() and not l.strip().startswith(("#", "//", "`"))), "")
    return difflib.SequenceMatcher(None, line.strip(), gold.strip()).ratio()


def score(dataset: str, pred: str, answers: List[str], all_classes=None) -> float:
    kind = METRIC[dataset]
    if kind == "classification":
        pred = pred.lstrip("\n").split("\n")[0]
    best = 0.0
    for g in answers:
        if kind == "f1":
            s = f1_score(pred, g)
        elif kind == "rouge":
            s = rouge_l(pred, g)
        elif kind == "classification":
            s = classification_score(pred, g, all_classes or [])
        elif kind == "count":
            s = count_score(pred, g)
        else:
            s = code_sim_score(pred, g)
        best = max(best, s)
    return best


# ------------------------------------------------------------------ prediction
def truncate_middle(tokenizer, prompt: str, max_length: int) -> str:
    ids = tokenizer(prompt, truncation=False).input_ids
    if len(ids) <= max_length:
        return prompt
    half = max_length // 2
    return tokenizer.decode(ids[:half], skip_special_tokens=True) + tokenizer.decode(ids[-half:], skip_special_tokens=True)


def set_compress(capacity: int, window: int = 32, kernel: int = 7, pooling: str = "maxpool"):
    """capacity 0 = full KV."""
    from bigdl_amd.transformers import kv
    if capacity:
        os.environ["IPEX_LLM_COMPRESS_KV_CACHE"] = "1"
        kv.CompressLayer.max_capacity_prompt = capacity
        kv.CompressLayer.window_size, kv.CompressLayer.kernel_size, kv.CompressLayer.pooling = window, kernel, pooling
    else:
        os.environ.pop("IPEX_LLM_COMPRESS_KV_CACHE", None)


def run_dataset(model, tokenizer, rows: List[dict], dataset: str, max_length: int, device) -> Dict[str, float]:
    import torch
    total = 0.0
    for r in rows:
        prompt = truncate_middle(tokenizer, PROMPTS[dataset].format(**r), max_length)
        ids = tokenizer(prompt, return_tensors="pt").input_ids.to(device)
        with torch.no_grad():
            out = model._hf_generate(ids, max_new_tokens=MAX_GEN[dataset], do_sample=False) \
                if os.environ.get("IPEX_LLM_COMPRESS_KV_CACHE") == "1" and hasattr(model, "_hf_generate") \
                else model.generate(ids, max_new_tokens=MAX_GEN[dataset], do_sample=False)
        pred = tokenizer.decode(out[0, ids.shape[1]:], skip_special_tokens=True)
        total += score(dataset, pred, r["answers"], r.get("all_classes"))
    return {"n": len(rows), "score": 100.0 * total / max(1, len(rows))}


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True)
    ap.add_argument("--data-dir", required=True)
    ap.add_argument("--datasets", nargs="+", default=list(PROMPTS))
    ap.add_argument("--compress", nargs="*", default=["c512", "c1024"], choices=list(COMPRESS))
    ap.add_argument("--full-kv", action="store_true")
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--max-length", type=int, default=3500)
    ap.add_argument("--limit", type=int, default=0)
    a = ap.parse_args()
    from transformers import AutoTokenizer
    from ipex_llm.transformers import AutoModelForCausalLM
    device = "cuda" if torch.cuda.is_available() else "cpu"
    model = AutoModelForCausalLM.from_pretrained(a.model, load_in_low_bit=a.low_bit, trust_remote_code=True).eval()
    if device != "cpu":
        model = model.half().to(device)
    tok = AutoTokenizer.from_pretrained(a.model, trust_remote_code=True)
    settings = ([("full_kv", 0)] if a.full_kv else []) + [(c, COMPRESS[c]) for c in a.compress]
    for name, cap in settings:
        set_compress(cap)
        for ds in a.datasets:
            rows = [json.loads(l) for l in open(os.path.join(a.data_dir, f"{ds}.jsonl")) if l.strip()]
            rows = rows[: a.limit] if a.limit else rows
            print(json.dumps({"setting": name, "dataset": ds, **run_dataset(model, tok, rows, ds, a.max_length, device)}))


if __name__ == "__main__":
    main()
