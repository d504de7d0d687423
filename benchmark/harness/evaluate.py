"""Multiple-choice / log-likelihood evaluation of a low-bit model (ARC, HellaSwag, Winogrande, TruthfulQA-MC, MMLU style).

The reference plugs its ``AutoModelForCausalLM`` into EleutherAI's lm-evaluation-harness
(python/llm/dev/benchmark/harness/ipexllm.py:32-59, ``run_llb.py``) and compares with stored fp16 baselines
(python/llm/test/benchmark/harness/fp16.csv).  Two entry points here:

* ``BigdlLM``: the same adapter -- ``lm_eval --model bigdl-llm --model_args pretrained=<path>,load_in_low_bit=sym_int4``
  when lm-eval is installed (registered on import of this module);
* ``python benchmark/harness/evaluate.py --model <path> --task-file arc.jsonl``: a dependency-free scorer for boxes
  without network access.  Task file: one JSON object per line, ``{"context": str, "choices": [str, ...], "gold": int}``;
  a choice's score is the sum of its token log-probabilities given the context (``acc``), or that sum divided by the
  choice's length in characters (``acc_norm``), exactly the two numbers the harness reports.
"""
from __future__ import annotations

import argparse
import json
from typing import List, Sequence

import torch


@torch.no_grad()
def loglikelihood(model, tokenizer, context: str, continuation: str, device=None, max_length: int = 2048) -> float:
    """Sum of log p(continuation tokens | context, earlier continuation tokens)."""
    ctx = tokenizer(context, add_special_tokens=True).input_ids if context else [tokenizer.bos_token_id or 0]
    cont = tokenizer(continuation, add_special_tokens=False).input_ids
    ids = (ctx + cont)[-max_length:]
    n_cont = min(len(cont), len(ids) - 1)
    t = torch.tensor([ids], device=device)
    out = model(t)
    logits = out.logits if hasattr(out, "logits") else out[0]
    if logits.shape[1] != len(ids):
        raise RuntimeError("the model returned logits for the last position only; evaluate with IPEX_LLM_LAST_LM_HEAD=0")
    lp = torch.log_softmax(logits[0, -n_cont - 1:-1].float(), -1)
    tgt = t[0, -n_cont:]
    return float(lp.gather(-1, tgt[:, None]).sum())


def score_task(model, tokenizer, rows: Sequence[dict], device=None) -> dict:
    acc = acc_norm = 0
    for r in rows:
        lls = [loglikelihood(model, tokenizer, r["context"], c, device) for c in r["choices"]]
        norm = [ll / max(1, len(c)) for ll, c in zip(lls, r["choices"])]
        acc += int(max(range(len(lls)), key=lls.__getitem__) == r["gold"])
        acc_norm += int(max(range(len(norm)), key=norm.__getitem__) == r["gold"])
    n = max(1, len(rows))
    return {"n": len(rows), "acc": acc / n, "acc_norm": acc_norm / n}


def load_rows(path: str, limit: int = 0) -> List[dict]:
    rows = [json.loads(l) for l in open(path) if l.strip()]
    return rows[:limit] if limit else rows


def register_lm_eval():
    """Register the ``bigdl-llm`` model type with lm-evaluation-harness if it is importable."""
    try:
        from lm_eval.api.registry import register_model
        from lm_eval.models.huggingface import HFLM
    except ImportError:
        return None
    from ipex_llm.transformers import AutoModelForCausalLM

    @register_model("bigdl-llm", "ipex-llm")
    class BigdlLM(HFLM):
        AUTO_MODEL_CLASS = AutoModelForCausalLM

        def __init__(self, *args, load_in_low_bit="sym_int4", **kwargs):
            import os
            os.environ.setdefault("IPEX_LLM_LAST_LM_HEAD", "0")       # log-likelihoods need every position
            self._low_bit = load_in_low_bit
            super().__init__(*args, **kwargs)

        def _create_model(self, pretrained, **kwargs):
            kwargs.pop("parallelize", None)
            self._model = AutoModelForCausalLM.from_pretrained(pretrained, load_in_low_bit=self._low_bit,
                                                               trust_remote_code=kwargs.get("trust_remote_code", True))
            if torch.cuda.is_available():
                self._model = self._model.half().to("cuda")
    return BigdlLM


BigdlLM = register_lm_eval()


def main():
    import os
    os.environ.setdefault("IPEX_LLM_LAST_LM_HEAD", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True)
    ap.add_argument("--task-file", required=True, nargs="+")
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--limit", type=int, default=0)
    a = ap.parse_args()
    from transformers import AutoTokenizer
    from ipex_llm.transformers import AutoModelForCausalLM
    model = AutoModelForCausalLM.from_pretrained(a.model, load_in_low_bit=a.low_bit, trust_remote_code=True).eval()
    if a.device != "cpu":
        model = model.half().to(a.device)
    tok = AutoTokenizer.from_pretrained(a.model, trust_remote_code=True)
    for f in a.task_file:
        res = score_task(model, tok, load_rows(f, a.limit), a.device)
        print(json.dumps({"task": os.path.basename(f), "model": a.model, "low_bit": a.low_bit, **res}))


if __name__ == "__main__":
    main()
