"""Prompt-lookup decoding: draft tokens come from n-gram matches in the text seen so far, the model verifies
a whole draft in one multi-token step.

Capability map to the reference python/llm/src/ipex_llm/transformers/lookup.py: the ``generate(lookahead=...)``
front door and performance-mode auto-enable :49-136, ``PromptLookupCandidateGenerator`` :145-270,
``lookup_generate`` :273-470.  On the MI355X engine a verification step of n tokens is one hipGraph replay whose
cost is almost the cost of a single-token step (the weights are streamed once either way), and a rejected tail is
undone by writing the KV length counter.
"""
from __future__ import annotations

import os
import time
from typing import Optional

import torch

from .stepper import make_stepper

PERFORMANCE_MODE_LOOKUP_INPUT_THRESHOLD = 100
SPEC_ONLY_KWARGS = ("max_step_draft", "th_stop_draft", "hf_adjust", "auto_th_stop_draft", "auto_parameters",
                    "min_step_draft", "th_batch_num")


def performance_mode_lookahead(input_shape, is_embeds=False) -> Optional[int]:
    """IPEX_LLM_PERFORMANCE_MODE=1 (or BIGDL_PERFORMANCE_MODE=1) turns lookup on for prompts of >= 100 tokens."""
    flag = os.environ.get("IPEX_LLM_PERFORMANCE_MODE", os.environ.get("BIGDL_PERFORMANCE_MODE"))
    if flag == "1" and input_shape is not None and not is_embeds and \
            input_shape[1] >= PERFORMANCE_MODE_LOOKUP_INPUT_THRESHOLD:
        return 2
    return None


class PromptLookupCandidateGenerator:
    """Incremental n-gram index over the running token sequence.

    ``table[ngram tuple] -> list of positions right after an occurrence`` is extended as tokens are accepted, so a
    lookup costs O(max_matching_ngram_size) dictionary probes instead of a sliding-window scan of the sequence.
    The most recent earlier occurrence wins (it is the likeliest continuation in copy-heavy prompts).
    """

    def __init__(self, num_output_tokens: int = 10, max_matching_ngram_size: Optional[int] = None,
                 device: str = "cpu"):
        self.num_output_tokens = num_output_tokens
        self.max_matching_ngram_size = max_matching_ngram_size or 2
        self.max_candidates = 9
        self.min_candidates = 0
        self.tokens: list = []
        self.table: dict = {}
        self.indexed = 0          # ngrams ending before this position are in the table
        if self.max_matching_ngram_size <= 0 or num_output_tokens <= 0:
            raise ValueError("Invalid max_matching_ngram_size or num_output_tokens")

    def init_look_up_table(self, input_ids):
        self.tokens = input_ids.view(-1).tolist() if isinstance(input_ids, torch.Tensor) else list(input_ids)
        self.table.clear()
        self.indexed = 0
        self._index()

    def _index(self):
        toks, n_max = self.tokens, self.max_matching_ngram_size
        # an ngram ending at position e (exclusive) is indexed once a continuation token exists (e < len)
        for e in range(max(self.indexed, 1), len(toks)):
            for n in range(1, n_max + 1):
                if e - n < 0:
                    break
                self.table.setdefault(tuple(toks[e - n:e]), []).append(e)
        self.indexed = max(self.indexed, len(toks))

    def update_look_up_table(self, new_ids):
        self.tokens.extend(new_ids.view(-1).tolist() if isinstance(new_ids, torch.Tensor) else list(new_ids))
        self._index()

    def get_candidates(self, input_ids=None):
        """-> list of draft tokens (possibly empty) continuing the current sequence."""
        if self.num_output_tokens <= 0:
            return []
        toks = self.tokens
        for n in range(min(self.max_matching_ngram_size, len(toks)), 0, -1):
            hits = self.table.get(tuple(toks[-n:]))
            if hits:
                start = hits[-1]
                cand = toks[start:start + self.num_output_tokens]
                if cand:
                    return cand
        return []

    def update_candidate_strategy(self, candidate_num: int, num_matches: int, accept_rate: float = 0.0):
        """Draft longer after a fully accepted draft, shorter after a mostly rejected one."""
        if num_matches == self.num_output_tokens:
            self.num_output_tokens = min(self.num_output_tokens + 1, self.max_candidates)
        elif candidate_num > num_matches:
            self.num_output_tokens = max(self.num_output_tokens - 1, self.min_candidates + 1)


def _pick(logits, do_sample, temperature, top_k, top_p):
    from ..models.decoder import sample_logits
    if do_sample:
        shp = logits.shape[:-1]
        return sample_logits(logits.reshape(-1, logits.shape[-1]), temperature, top_k, top_p).view(shp)
    return logits.argmax(-1)


@torch.no_grad()
def lookup_generate(self, inputs=None, max_new_tokens: int = 10, num_output_tokens: int = 10,
                    max_matching_ngram_size: Optional[int] = None, generation_config=None, streamer=None,
                    attention_mask=None, **sampling_kwargs):
    """Greedy (or sampled) generation with prompt-lookup drafts; batch size 1.

    Sets ``n_token_generated``, ``e2e_time_without_first``, ``first_token_time``, ``first_cost``, ``rest_cost_mean``,
    ``n_matched``, ``n_drafted`` on the model.
    """
    input_ids = sampling_kwargs.pop("input_ids", inputs)
    assert input_ids is not None and input_ids.shape[0] == 1, "prompt lookup decoding supports batch size 1"
    gc = generation_config or getattr(self, "generation_config", None)
    do_sample = bool(sampling_kwargs.get("do_sample", False))
    temperature = sampling_kwargs.get("temperature", 1.0) or 1.0
    top_k = sampling_kwargs.get("top_k", 0) or 0
    top_p = sampling_kwargs.get("top_p", 1.0) or 1.0
    eos = sampling_kwargs.get("eos_token_id", getattr(gc, "eos_token_id", None))
    eos = set([eos] if isinstance(eos, int) else (eos or []))
    stopping_criteria = sampling_kwargs.get("stopping_criteria")

    stepper = sampling_kwargs.pop("stepper", None) or make_stepper(
        self, capacity=input_ids.shape[1] + max_new_tokens + num_output_tokens + 16)
    dev = stepper.device
    cg = PromptLookupCandidateGenerator(num_output_tokens, max_matching_ngram_size)
    cg.init_look_up_table(input_ids)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    sync()
    tic = time.perf_counter()
    logits = stepper.prefill(input_ids)
    cur = _pick(logits, do_sample, temperature, top_k, top_p).view(1, 1)   # first generated token, not yet in the cache
    sync()
    first_time = time.perf_counter() - tic
    generated = [int(cur)]
    cg.update_look_up_table(cur)
    if streamer is not None:
        streamer.put(input_ids.cpu())
        streamer.put(cur.cpu())
    n_matched = n_drafted = 0
    tic = time.perf_counter()
    step_times = []
    done = int(cur) in eos
    while not done and len(generated) < max_new_tokens:
        t0 = time.perf_counter()
        room = max_new_tokens - len(generated)
        cand = cg.get_candidates()[: max(0, room - 1)]
        tokens = torch.tensor([[generated[-1]] + cand], dtype=torch.long, device=dev)
        base = stepper.length
        out = stepper.forward(tokens)                                   # [1, 1+len(cand), V]
        choice = _pick(out[0], do_sample, temperature, top_k, top_p).tolist()
        m = 0
        while m < len(cand) and cand[m] == choice[m]:
            m += 1
        accepted = choice[: m + 1]                                      # m verified drafts + the model's next token
        n_drafted += len(cand)
        n_matched += m
        stepper.crop(base + 1 + m)
        if eos:
            for i, t in enumerate(accepted):
                if t in eos:
                    accepted = accepted[: i + 1]
                    done = True
                    break
        generated.extend(accepted)
        new = torch.tensor(accepted, dtype=torch.long)
        cg.update_look_up_table(new)
        cg.update_candidate_strategy(len(cand), m)
        if streamer is not None:
            streamer.put(new)
        if stopping_criteria is not None:
            full = torch.cat([input_ids.cpu(), torch.tensor([generated])], 1)
            if bool(torch.as_tensor(stopping_criteria(full, None)).all()):
                done = True
        sync()
        step_times.append(time.perf_counter() - t0)
    sync()
    rest = time.perf_counter() - tic
    if streamer is not None:
        streamer.end()
    generated = generated[:max_new_tokens]
    stats = dict(n_token_generated=len(generated), e2e_time_without_first=rest, first_token_time=first_time,
                 first_cost=first_time, rest_cost_mean=rest / max(1, len(generated) - 1), n_matched=n_matched,
                 n_drafted=n_drafted, lookup_step_times=step_times)
    for k, v in stats.items():
        object.__setattr__(self, k, v)
    return torch.cat([input_ids.to(dev), torch.tensor([generated], dtype=input_ids.dtype, device=dev)], 1)
