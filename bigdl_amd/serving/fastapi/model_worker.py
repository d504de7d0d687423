"""Single-GPU serving worker: a request queue drained by one generation loop.

Reference: python/llm/src/ipex_llm/serving/fastapi/model_worker.py:28-210 (``ModelWorker.add_request`` /
``process_step``: one ``generate`` thread per admitted request with a ``TextIteratorStreamer``).

Here generation runs on the worker's executor thread through ``model.generate`` (the hipGraph decode engine for
supported families), one request at a time per GPU stream -- decode at batch 1 is weight-bandwidth bound, so a
second concurrent ``generate`` on the same GPU would only interleave with it -- and text deltas are handed to the
event loop through an ``asyncio.Queue`` per request.  Requests wait in FIFO order; ``max_queue`` bounds the backlog.
"""
from __future__ import annotations

import asyncio
import threading
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch


@dataclass
class Request:
    request_id: str
    prompt: str
    max_new_tokens: int = 32
    min_new_tokens: int = 0
    do_sample: bool = False
    temperature: Optional[float] = None
    top_p: Optional[float] = None
    top_k: Optional[int] = None
    repetition_penalty: Optional[float] = None
    stop: List[str] = field(default_factory=list)
    input_ids: Optional[List[int]] = None
    arrival: float = field(default_factory=time.time)


class _QueueStreamer:
    """Streamer protocol of ``generate`` -> (delta text, finished) items on an asyncio queue (thread-safe)."""

    def __init__(self, tokenizer, loop, queue, skip_prompt=True):
        self.tokenizer, self.loop, self.queue = tokenizer, loop, queue
        self.skip_prompt, self.first = skip_prompt, True
        self.tokens: List[int] = []
        self.printed = 0
        self.n_generated = 0

    def _emit(self, item):
        self.loop.call_soon_threadsafe(self.queue.put_nowait, item)

    def put(self, value):
        if self.skip_prompt and self.first:
            self.first = False
            return
        ids = value.reshape(-1).tolist() if isinstance(value, torch.Tensor) else list(value)
        self.tokens.extend(ids)
        self.n_generated += len(ids)
        text = self.tokenizer.decode(self.tokens, skip_special_tokens=True)
        # hold back a trailing partial character / word piece
        if text.endswith("�"):
            return
        delta = text[self.printed:]
        if delta:
            self.printed = len(text)
            self._emit((delta, False))

    def end(self):
        text = self.tokenizer.decode(self.tokens, skip_special_tokens=True)
        self._emit((text[self.printed:], True))

    def fail(self, message: str):
        """Close the stream with an error note instead of leaving the client to its token timeout."""
        self._emit((message, True))


class ModelWorker:
    def __init__(self, checkpoint=None, low_bit="sym_int4", model_type="normal", torch_dtype=torch.float16, model=None,
                 tokenizer=None, device=None, max_queue: int = 256):
        self.dtype = torch_dtype
        self.model_type = model_type
        self.device = device or ("cuda" if torch.cuda.is_available() else "cpu")
        if model is None:
            model, tokenizer = self.load_model(checkpoint, low_bit, model_type)
        self.model, self.tokenizer = model, tokenizer
        self.waiting_requests: asyncio.Queue = asyncio.Queue(maxsize=max_queue)
        self.streamer: Dict[str, asyncio.Queue] = {}
        self.stats = dict(requests=0, prompt_tokens=0, generated_tokens=0, busy_s=0.0, first_token_s=[], per_token_s=[])
        self._lock = threading.Lock()

    def load_model(self, model_path, low_bit="sym_int4", model_type="normal"):
        from transformers import AutoTokenizer
        from ...transformers import AutoModelForCausalLM
        try:
            model = AutoModelForCausalLM.load_low_bit(model_path, torch_dtype=self.dtype)
        except Exception:
            model = AutoModelForCausalLM.from_pretrained(model_path, load_in_low_bit=low_bit, torch_dtype=self.dtype,
                                                         use_cache=True)
        if self.device != "cpu":
            model = model.to(self.dtype).to(self.device)
        tokenizer = AutoTokenizer.from_pretrained(model_path)
        return model.eval(), tokenizer

    async def add_request(self, req: Request) -> asyncio.Queue:
        q: asyncio.Queue = asyncio.Queue()
        self.streamer[req.request_id] = q
        await self.waiting_requests.put(req)
        return q

    def _generate(self, req: Request, loop):
        tok = self.tokenizer
        if req.input_ids is not None:
            ids = torch.tensor([req.input_ids])
        else:
            ids = tok(req.prompt, return_tensors="pt").input_ids
        dev = next(p for p in self.model.parameters() if p.numel() > 0).device
        ids = ids.to(dev)
        streamer = _QueueStreamer(tok, loop, self.streamer[req.request_id])
        kw = dict(max_new_tokens=req.max_new_tokens, streamer=streamer)
        if req.min_new_tokens:
            kw["min_new_tokens"] = req.min_new_tokens
        sample = req.do_sample or (req.temperature is not None and req.temperature > 0 and req.temperature != 1.0) \
            or req.top_p is not None or req.top_k is not None
        if sample:
            kw.update(do_sample=True)
            for k in ("temperature", "top_p", "top_k"):
                if getattr(req, k) is not None:
                    kw[k] = getattr(req, k)
        else:
            kw["do_sample"] = False
        if req.repetition_penalty:
            kw["repetition_penalty"] = req.repetition_penalty
        t0 = time.time()
        try:
            with torch.inference_mode():
                self.model.generate(ids, **kw)
        except Exception as e:                      # surface the failure to the waiting client instead of hanging it
            loop.call_soon_threadsafe(self.streamer[req.request_id].put_nowait, (f"[generation failed: {e}]", True))
        dt = time.time() - t0
        with self._lock:
            s = self.stats
            s["requests"] += 1
            s["prompt_tokens"] += ids.shape[1]
            s["generated_tokens"] += streamer.n_generated
            s["busy_s"] += dt
            fc, rc = getattr(self.model, "first_cost", None), getattr(self.model, "rest_cost_mean", None)
            if fc:
                s["first_token_s"] = (s["first_token_s"] + [float(fc)])[-1000:]
            if rc:
                s["per_token_s"] = (s["per_token_s"] + [float(rc)])[-1000:]

    async def process_step(self):
        """Take the next waiting request and run it to completion on the executor thread."""
        req = await self.waiting_requests.get()
        loop = asyncio.get_running_loop()
        await loop.run_in_executor(None, self._generate, req, loop)
        return req.request_id

    def finish(self, request_id: str):
        self.streamer.pop(request_id, None)


class BatchedModelWorker(ModelWorker):
    """Serving worker with iteration-level batching: up to ``max_num_seqs`` requests decode together in the slots of
    one native-engine KV cache (``bigdl_amd/serving/continuous.py``); requests are admitted between decode steps.
    Takes the place of the reference's vLLM-backed serving for families the native engine covers; other models fall
    back to the one-request-at-a-time loop of ``ModelWorker``."""

    def __init__(self, *args, max_num_seqs: int = 16, max_seq_len: int = 4096, max_prefilled_seqs: int = 1, batcher=None,
                 **kw):
        """``batcher``: a scheduler built by the caller instead of the one made here from the model's engine -- the
        source rank of a tensor-parallel group passes its ``serving/tp.py::TensorParallelBatcher``."""
        super().__init__(*args, **kw)
        from ...models.hf_patch import get_native_engine, native_engine_supported
        from ...transformers.utils import engine_kv_mode
        self.batcher = batcher
        if batcher is not None:
            batcher.on_token = self._on_token
        elif self.device != "cpu" and native_engine_supported(self.model):
            from ..continuous import ContinuousBatcher
            self.batcher = ContinuousBatcher(get_native_engine(self.model), max_num_seqs=max_num_seqs,
                                             max_seq_len=max_seq_len, kv_fp8=engine_kv_mode(),
                                             max_prefill_per_step=max_prefilled_seqs, on_token=self._on_token)
        self._loop = None
        self._text: Dict[str, _QueueStreamer] = {}

    def _on_token(self, seq, token):
        st = self._text.get(seq.request_id)
        if st is None:
            return
        st.put([token])
        if seq.finished:
            st.end()
            self._text.pop(seq.request_id, None)
            with self._lock:
                s = self.stats
                s["requests"] += 1
                s["prompt_tokens"] += len(seq.input_ids)
                s["generated_tokens"] += len(seq.output_ids)
                if seq.first_token_time is not None:
                    s["first_token_s"] = (s["first_token_s"] + [seq.first_token_time - seq.arrival])[-1000:]
                    if len(seq.output_ids) > 1:
                        s["per_token_s"] = (s["per_token_s"] + [(seq.finish_time - seq.first_token_time)
                                                                / (len(seq.output_ids) - 1)])[-1000:]

    def _admit(self, req: Request, loop):
        from ..continuous import SeqRequest
        ids = req.input_ids if req.input_ids is not None else self.tokenizer(req.prompt).input_ids
        st = _QueueStreamer(self.tokenizer, loop, self.streamer[req.request_id], skip_prompt=False)
        self._text[req.request_id] = st
        sample = req.do_sample or (req.temperature is not None and req.temperature > 0 and req.temperature != 1.0) \
            or req.top_p is not None or req.top_k is not None
        eos = getattr(self.model.generation_config, "eos_token_id", None) or self.tokenizer.eos_token_id
        try:
            self.batcher.submit(SeqRequest(req.request_id, list(ids), max_new_tokens=req.max_new_tokens,
                                           min_new_tokens=req.min_new_tokens, eos_token_id=eos, do_sample=bool(sample),
                                           temperature=req.temperature or 1.0, top_k=req.top_k or 0, top_p=req.top_p or 1.0))
        except ValueError as e:
            self._text.pop(req.request_id, None)
            loop.call_soon_threadsafe(self.streamer[req.request_id].put_nowait, (f"[generation failed: {e}]", True))

    def _iterate(self):
        t0 = time.time()
        try:
            with torch.inference_mode():
                self.batcher.step()
        except Exception as e:
            # a failed iteration (capacity assert, device error): end the requests it carried instead of leaving them
            # to the client-side token timeout with their slots held, then let the serving loop see the error
            for seq in self.batcher.fail_all("error"):
                st = self._text.pop(seq.request_id, None)
                if st is not None:
                    st.fail(f"[generation failed: {type(e).__name__}: {e}]")
            raise
        finally:
            with self._lock:
                self.stats["busy_s"] += time.time() - t0

    async def process_step(self):
        if self.batcher is None:
            return await super().process_step()
        loop = asyncio.get_running_loop()
        if not self.batcher.has_work():
            beat = getattr(self.batcher, "heartbeat_s", None)
            if beat:
                # a scheduler with followers (tensor-parallel ranks blocked in a host receive): an empty step every
                # ``heartbeat_s`` keeps their receive from timing out while no request arrives
                try:
                    self._admit(await asyncio.wait_for(self.waiting_requests.get(), timeout=beat), loop)
                except asyncio.TimeoutError:
                    pass
            else:
                self._admit(await self.waiting_requests.get(), loop)    # idle: block until a request arrives
        while not self.waiting_requests.empty():
            self._admit(self.waiting_requests.get_nowait(), loop)
        await loop.run_in_executor(None, self._iterate)
        return None
