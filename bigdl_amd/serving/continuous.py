"""Continuous batching on the native decoder engine.

The reference gets iteration-level batching from two places: its vLLM integration
(python/llm/src/ipex_llm/vllm/xpu/engine/engine.py:40-246) and the pipeline-parallel serving worker that keeps
``max_num_seqs`` sequences per batch in flight with partial prefill
(python/llm/src/ipex_llm/transformers/pipeline_parallel.py:482-521, 621-700, 807-927).  On one MI355X the same
effect falls out of the engine's fixed-capacity KV cache: every batch row is a *slot* with its own length counter
on the device, a new request is prefilled straight into a free slot, and all occupied slots advance together with
one hipGraph replay per token.  Batch-1 decode leaves the GPU latency bound (about 1.7 ms / token for a 7B int4
model), the same graph at 32 slots costs little more, so throughput scales with occupancy.

``ContinuousBatcher`` is the scheduler: ``submit`` queues a request, ``step`` admits waiting requests into free
slots (at most ``max_prefill_per_step`` per step, the reference's ``max_prefilled_seqs``), runs one decode step over
all slots and returns the new token of every running request.  When sequences are running while a prompt is
admitted, both go through one pass (``NativeDecoder.prefill(with_decode=True)``): the decode rows are appended to the
prompt's rows, so the step costs one read of the weights instead of two.  Prompts longer than ``max_prefill_tokens``
are admitted in chunks of that many tokens, one chunk per step (the reference's partial prefill,
python/llm/src/ipex_llm/transformers/pipeline_parallel.py:621-700), so a long prompt delays the running sequences by
one chunk's time per token instead of stalling them for the whole prompt.
"""
from __future__ import annotations

import collections
import os
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import torch


@dataclass
class SeqRequest:
    request_id: str
    input_ids: List[int]
    max_new_tokens: int = 32
    min_new_tokens: int = 0
    eos_token_id: Optional[List[int]] = None
    do_sample: bool = False
    temperature: float = 1.0
    top_k: int = 0
    top_p: float = 1.0
    # filled in by the scheduler
    output_ids: List[int] = field(default_factory=list)
    slot: int = -1
    finished: bool = False
    finish_reason: Optional[str] = None
    arrival: float = field(default_factory=time.perf_counter)
    first_token_time: Optional[float] = None
    finish_time: Optional[float] = None


class ContinuousBatcher:
    """Iteration-level scheduler over ``max_num_seqs`` cache slots of one ``NativeDecoder``."""

    def __init__(self, engine, max_num_seqs: int = 8, max_seq_len: int = 4096, kv_fp8: bool = False,
                 max_prefill_per_step: int = 8, on_token: Optional[Callable[[SeqRequest, int], None]] = None,
                 piggyback: Optional[bool] = None, dense_prefill_cache: Optional[bool] = None,
                 max_prefill_tokens: Optional[int] = 4096):
        self.engine = engine
        self.max_num_seqs = max_num_seqs
        self.max_seq_len = max_seq_len
        self.max_prefill_per_step = max(1, max_prefill_per_step)
        self.on_token = on_token
        self.waiting: collections.deque = collections.deque()
        self.running: Dict[int, SeqRequest] = {}
        self.free_slots = list(range(max_num_seqs))[::-1]
        cache = engine.ensure_cache(max_num_seqs, max_seq_len, fp8=kv_fp8)
        cache.enable_slots()
        self.cache = cache
        # prompt passes run on the fused low-bit prompt GEMM like everything else.  Opt-in (argument, or
        # BIGDL_PREFILL_DENSE_CACHE=1): keep fp16 copies of the weights and hand the prompt GEMMs to the library
        # (+2 bytes per weight of HBM; decode keeps streaming the low-bit tiles either way)
        if dense_prefill_cache is None:
            dense_prefill_cache = os.environ.get("BIGDL_PREFILL_DENSE_CACHE") == "1"
        if dense_prefill_cache:
            from ..models.decoder import Lin
            Lin.cache_dense = True
        self._tokens = None      # the engine's static [B, 1] token buffer (input of the next decode step)
        self.steps = 0
        self.generated = 0
        # decode rows ride on the prompt pass of a newly admitted request (BIGDL_SERVING_PIGGYBACK=0 turns it off)
        self.piggyback = os.environ.get("BIGDL_SERVING_PIGGYBACK", "1") != "0" if piggyback is None else piggyback
        self.mixed_steps = 0
        self.packed_prompts = 0
        # prompts longer than this go through several passes (chunked prefill), so that the running sequences keep
        # producing a token per pass instead of waiting for a whole long prompt; None / 0: one pass per prompt
        self.max_prefill_tokens = max_prefill_tokens
        self.packed_mixed_steps = 0
        # BIGDL_SERVING_PACKED_MIXED=1: packed admissions also carry the decode rows (_packed_mixed_pass).  Measured neutral
        # (Llama-2-7B, 64 slots: 3999 against 4061 tokens/s; Llama-3-8B shape: 5304 / 6368 against 5229 / 6433 at 64 / 128
        # slots): a 4096-row pass is compute bound, the weight read it saves is 1 % of it.  Off by default.
        self.packed_mixed = os.environ.get("BIGDL_SERVING_PACKED_MIXED", "0") == "1"
        self._prefilling = None                        # [request, tokens of its prompt already in the cache]

    # ------------------------------------------------------------------ queue
    def submit(self, req: SeqRequest) -> SeqRequest:
        if len(req.input_ids) == 0:
            raise ValueError(f"request {req.request_id}: empty prompt")
        if len(req.input_ids) + req.max_new_tokens > self.cache.capacity:
            raise ValueError(f"request {req.request_id}: prompt + max_new_tokens exceeds the slot capacity "
                             f"({self.cache.capacity})")
        if isinstance(req.eos_token_id, int):
            req.eos_token_id = [req.eos_token_id]
        self.waiting.append(req)
        return req

    def has_work(self) -> bool:
        return bool(self.waiting or self.running or self._prefilling)

    def fail_all(self, reason: str = "error"):
        """After an exception inside ``step``: finish every admitted request (running, being prefilled) with
        ``finish_reason = reason``, free their slots and leave the batcher ready for new work.  Returns the failed
        requests; waiting requests stay queued.  Without this a failing step leaves ``has_work()`` true for ever."""
        failed = list(self.running.values())
        if self._prefilling is not None:
            failed.append(self._prefilling[0])
            self._prefilling = None
        now = time.perf_counter()
        for req in failed:
            req.finished, req.finish_reason, req.finish_time = True, reason, now
            if req.slot is not None:
                self.cache.set_active(req.slot, False)
                self.running.pop(req.slot, None)
                if req.slot not in self.free_slots:
                    self.free_slots.append(req.slot)
        return failed

    # ------------------------------------------------------------------ one iteration
    def _token_buffer(self):
        if self._tokens is None:
            _, st = self.engine._get_graph(self.max_num_seqs, 1, True)
            self._tokens = st["tokens"]
        return self._tokens

    def _pick(self, req: SeqRequest, logits_row: torch.Tensor) -> int:
        if req.do_sample:
            from ..models.decoder import sample_logits
            return int(sample_logits(logits_row[None].float(), req.temperature, req.top_k, req.top_p)[0])
        return int(logits_row.argmax(-1))

    def _emit(self, req: SeqRequest, token: int, now: float, out: Dict[str, int]):
        req.output_ids.append(token)
        out[req.request_id] = token
        self.generated += 1
        if req.first_token_time is None:
            req.first_token_time = now
        n = len(req.output_ids)
        if req.eos_token_id and token in req.eos_token_id and n >= req.min_new_tokens:
            req.finished, req.finish_reason = True, "stop"
        elif n >= req.max_new_tokens:
            req.finished, req.finish_reason = True, "length"
        if self.on_token is not None:
            self.on_token(req, token)
        if req.finished:
            req.finish_time = now
            self.cache.set_active(req.slot, False)
            self.running.pop(req.slot, None)
            self.free_slots.append(req.slot)

    def _admit(self, req: SeqRequest):
        slot = self.free_slots.pop()
        req.slot = slot
        self.cache.set_active(slot, False)            # clears the slot's length counters
        self._prefilling = [req, 0]

    def _started(self, req: SeqRequest, logits_row, tokens, out):
        tok = self._pick(req, logits_row)
        tokens[req.slot, 0] = tok
        self.cache.set_active(req.slot, True)
        self.running[req.slot] = req
        self._emit(req, tok, time.perf_counter(), out)

    def _advance(self, logits, nxt, tokens, out):
        host = nxt.tolist()                           # one device sync per iteration
        now = time.perf_counter()
        for slot, req in list(self.running.items()):
            tok = host[slot]
            if req.do_sample:
                tok = self._pick(req, logits[slot])
                tokens[slot, 0] = tok
            self._emit(req, tok, now, out)

    def _prompt_pass(self, tokens, out, with_decode: bool) -> bool:
        """The next chunk of the prompt being admitted (at most ``max_prefill_tokens`` tokens), optionally with the
        decode rows of the running sequences on the same pass.  True when the prompt is complete."""
        eng = self.engine
        req, done = self._prefilling
        n = len(req.input_ids) - done
        if self.max_prefill_tokens:
            n = min(n, self.max_prefill_tokens)
        ids = torch.tensor([req.input_ids[done:done + n]], dtype=torch.int64, device=eng.device)
        if with_decode:
            first, logits, nxt = eng.prefill(ids, last_only=True, slot=req.slot, with_decode=True)
            self._advance(logits, nxt, tokens, out)
            self.mixed_steps += 1
        else:
            first = eng.prefill(ids, last_only=True, slot=req.slot)
        self._prefilling[1] = done + n
        if done + n < len(req.input_ids):
            return False
        self._prefilling = None
        self._started(req, first[0, -1], tokens, out)
        return True

    def _packed_mixed_pass(self, tokens, out) -> bool:
        """Two or more waiting whole prompts AND running sequences: one pass carries the packed prompts and one decode
        row per slot (``prefill_packed(with_decode=True)``), i.e. one read of the weights for the admissions and the step."""
        eng = self.engine
        if not (self.piggyback and self.packed_mixed and self.running and self._prefilling is None and len(self.waiting) >= 2
                and len(self.free_slots) >= 2 and getattr(eng, "packed_with_decode", False)):
            return False
        batch, budget = [], self.max_prefill_tokens or (1 << 30)
        while self.waiting and len(self.free_slots) > len(batch) and len(batch) < self.max_prefill_per_step:
            n = len(self.waiting[0].input_ids)
            if n > budget:
                break
            budget -= n
            batch.append(self.waiting.popleft())
        if len(batch) < 2:
            for req in reversed(batch):
                self.waiting.appendleft(req)
            return False
        for req in batch:
            req.slot = self.free_slots.pop()
            self.cache.set_active(req.slot, False)
        first, logits, nxt = eng.prefill_packed([r.input_ids for r in batch], [r.slot for r in batch], with_decode=True)
        self._advance(logits, nxt, tokens, out)          # the sequences that were running (before the new ones join)
        for j, req in enumerate(batch):
            self._started(req, first[j], tokens, out)
        self.packed_prompts += len(batch)
        self.mixed_steps += 1
        self.packed_mixed_steps += 1
        return True

    def step(self) -> Dict[str, int]:
        """Admit, decode once, return {request_id: new token}."""
        eng, out = self.engine, {}
        tokens = self._token_buffer()
        admitted = 0
        if self._packed_mixed_pass(tokens, out):
            self.steps += 1
            return out
        if self._prefilling is None and self.waiting and self.free_slots:
            self._admit(self.waiting.popleft())
        if self._prefilling is not None and self.running and self.piggyback:
            # one pass for both: the running sequences' next token rides on the prompt chunk's GEMMs
            admitted += self._prompt_pass(tokens, out, with_decode=True)
        else:
            if self.running:
                # the sequences that are already running advance by one token
                logits, nxt = eng.decode(None, 1, chain=True)
                self._advance(logits, nxt, tokens, out)
            if self._prefilling is not None:
                admitted += self._prompt_pass(tokens, out, with_decode=False)
        # further admissions of this step: whole prompts, packed into one pass (one read of the weights for all of
        # them) while they fit the token budget of a pass
        batch, budget = [], self.max_prefill_tokens or (1 << 30)
        while self._prefilling is None and self.waiting and len(self.free_slots) > len(batch) \
                and 0 < admitted + len(batch) < self.max_prefill_per_step and hasattr(eng, "prefill_packed"):
            n = len(self.waiting[0].input_ids)
            if n > budget:
                break
            budget -= n
            batch.append(self.waiting.popleft())
        if len(batch) == 1:
            self._admit(batch[0])
            admitted += self._prompt_pass(tokens, out, with_decode=False)
        elif batch:
            for req in batch:
                req.slot = self.free_slots.pop()
                self.cache.set_active(req.slot, False)
            logits = eng.prefill_packed([r.input_ids for r in batch], [r.slot for r in batch])
            for j, req in enumerate(batch):
                self._started(req, logits[j], tokens, out)
            admitted += len(batch)
            self.packed_prompts += len(batch)
        self.steps += 1
        return out

    def run_until_done(self, max_steps: int = 1 << 30) -> None:
        n = 0
        while self.has_work() and n < max_steps:
            self.step()
            n += 1


def generate_batch(engine, prompts: List[List[int]], max_new_tokens: int = 32, max_num_seqs: int = 8,
                   eos_token_id=None, kv_fp8: bool = False, **sampling) -> List[List[int]]:
    """Run ``prompts`` (token id lists of any lengths) through a ``ContinuousBatcher``; returns the new tokens of
    each prompt in order."""
    cap = max(len(p) for p in prompts) + max_new_tokens + 1
    cb = ContinuousBatcher(engine, max_num_seqs=min(max_num_seqs, len(prompts)), max_seq_len=cap, kv_fp8=kv_fp8,
                           max_prefill_per_step=max_num_seqs)
    reqs = [cb.submit(SeqRequest(str(i), list(p), max_new_tokens=max_new_tokens, eos_token_id=eos_token_id, **sampling))
            for i, p in enumerate(prompts)]
    cb.run_until_done()
    return [r.output_ids for r in reqs]
