"""Perplexity of a (low-bit) causal LM over a token stream with a sliding window.

Reference: python/llm/dev/benchmark/perplexity/ppl.py (``BigDLPPL.perplexity_hf``: windows of ``seq_len`` tokens with
stride, cross-entropy on the continuation part, ppl = exp(mean nll)).  Datasets are not downloadable on the test
boxes: pass ``--text-file`` (plain text) or ``--token-file`` (a ``torch.save``d 1-D LongTensor).
"""
from __future__ import annotations

import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))


class BigDLPPL:
    def __init__(self, model_path=None, device="cuda", low_bit="sym_int4", model=None, **model_kwargs):
        if model is None:
            from bigdl_amd.transformers import AutoModelForCausalLM
            model = AutoModelForCausalLM.from_pretrained(model_path, load_in_low_bit=low_bit, **model_kwargs)
            if device != "cpu":
                model = model.half().to(device)
        self.model = model.eval()
        self.device = next(p for p in model.parameters() if p.numel() > 0).device
        lm_head = getattr(self.model, "lm_head", None)
        if lm_head is not None and hasattr(lm_head, "optimize_lm_head"):
            lm_head.optimize_lm_head = False           # every position's logits are needed

    @torch.no_grad()
    def perplexity_hf(self, tokens: torch.Tensor, seq_len: int = 512, stride: int = None):
        stride = stride or seq_len
        tokens = tokens.view(-1)
        nll, count = 0.0, 0
        prev_end = 0
        for begin in range(0, max(1, tokens.numel() - 1), stride):
            end = min(begin + seq_len, tokens.numel())
            trg_len = end - prev_end
            ids = tokens[begin:end].unsqueeze(0).to(self.device)
            logits = self.model(ids).logits.float()
            shift_logits = logits[0, :-1]
            shift_labels = ids[0, 1:]
            keep = min(trg_len, shift_labels.numel())
            loss = torch.nn.functional.cross_entropy(shift_logits[-keep:], shift_labels[-keep:], reduction="sum")
            nll += loss.item()
            count += keep
            prev_end = end
            if end == tokens.numel():
                break
        return math.exp(nll / max(1, count))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_path", required=True)
    ap.add_argument("--precisions", nargs="+", default=["sym_int4"])
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--seq_len", type=int, default=512)
    ap.add_argument("--text-file")
    ap.add_argument("--token-file")
    a = ap.parse_args()
    if a.token_file:
        tokens = torch.load(a.token_file, weights_only=True)
    else:
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(a.model_path)
        tokens = tok(open(a.text_file).read(), return_tensors="pt").input_ids[0]
    summary = {}
    for p in a.precisions:
        summary[p] = BigDLPPL(a.model_path, a.device, p).perplexity_hf(tokens, a.seq_len)
        print(f"{p}: ppl = {summary[p]:.4f}")
    return summary


if __name__ == "__main__":
    main()
