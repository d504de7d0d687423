"""FastChat model worker with iteration-level batching.

Reference: python/llm/src/ipex_llm/serving/fastchat/vllm_worker.py (``VLLMWorker(BaseModelWorker)`` around the
IPEX-LLM vLLM engine: concurrent requests are batched by the engine, ``generate_stream`` yields NUL-terminated JSON
chunks ``{"text", "error_code", "usage", "cumulative_logprob", "finish_reason"}``, stop strings / stop token ids end a
request early and abort it in the engine, a background task aborts the request when the client goes away).

vLLM is not part of this stack: the batching engine here is the native engine's continuous batcher
(``serving/continuous.py`` behind ``serving/fastapi/model_worker.py::BatchedModelWorker``), which plays the same role --
every request that arrives while others decode joins the running batch at the next step.  The HTTP surface
(``/worker_generate_stream``, ``/worker_generate``, ``/worker_get_status``, ``/count_token``, ``/worker_get_conv_template``,
``/model_details``) is FastChat's worker protocol, so a FastChat controller / OpenAI front end drives it unchanged.
"""
from __future__ import annotations

import asyncio
import json
import uuid
from typing import List, Optional

import torch
from fastapi import BackgroundTasks, FastAPI
from fastapi import Request as HttpRequest           # (module level: the endpoint annotations below are resolved here)
from fastapi.responses import JSONResponse, StreamingResponse

from .ipex_llm_worker import HAVE_FASTCHAT, BaseModelWorker, worker_id  # noqa: F401
from ..fastapi.model_worker import BatchedModelWorker, Request


class VLLMWorker(BaseModelWorker):
    def __init__(self, controller_addr: str = "", worker_addr: str = "", worker_id: str = worker_id, model_path: str = "",
                 model_names: Optional[List[str]] = None, limit_worker_concurrency: int = 64, no_register: bool = True,
                 llm_engine: Optional[BatchedModelWorker] = None, conv_template: Optional[str] = None,
                 load_in_low_bit: str = "sym_int4", max_num_seqs: int = 16, max_seq_len: int = 4096, model=None, tokenizer=None):
        super().__init__(controller_addr, worker_addr, worker_id, model_path, model_names, limit_worker_concurrency,
                         conv_template)
        if llm_engine is None:
            kw = dict(model=model, tokenizer=tokenizer) if model is not None else dict(checkpoint=model_path, low_bit=load_in_low_bit)
            llm_engine = BatchedModelWorker(max_num_seqs=max_num_seqs, max_seq_len=max_seq_len, **kw)
        self.llm_engine = llm_engine
        self.tokenizer = llm_engine.tokenizer
        self.context_len = getattr(llm_engine.model.config, "max_position_embeddings", 2048)
        self._driver: Optional[asyncio.Task] = None
        self.call_ct = 0
        if not no_register:
            self.init_heart_beat()

    # ------------------------------------------------------------------ engine loop
    def _ensure_driver(self):
        """One task steps the engine for as long as the event loop lives (started by the first request)."""
        if self._driver is None or self._driver.done():
            async def drive():
                while True:
                    await self.llm_engine.process_step()
            self._driver = asyncio.get_running_loop().create_task(drive())

    def _abort(self, request_id: str):
        b = getattr(self.llm_engine, "batcher", None)
        if b is None:
            return
        if hasattr(b, "abort"):
            b.abort(request_id)
            return
        for req in list(b.waiting):
            if req.request_id == request_id:
                b.waiting.remove(req)
                req.finished, req.finish_reason = True, "abort"
        for slot, req in list(b.running.items()):
            if req.request_id == request_id:
                b.cache.set_active(slot, False)
                b.running.pop(slot, None)
                b.free_slots.append(slot)
                req.finished, req.finish_reason = True, "abort"
        self.llm_engine._text.pop(request_id, None)

    # ------------------------------------------------------------------ FastChat worker protocol
    async def generate_stream(self, params):
        self.call_ct += 1
        self._ensure_driver()
        prompt = params.pop("prompt")
        request_id = params.pop("request_id", None) or uuid.uuid4().hex
        temperature = float(params.get("temperature", 1.0))
        top_p = float(params.get("top_p", 1.0))
        top_k = int(params.get("top_k", -1))
        max_new_tokens = int(params.get("max_new_tokens", 256))
        echo = bool(params.get("echo", True))
        stop = set()
        stop_str = params.get("stop")
        if isinstance(stop_str, str) and stop_str:
            stop.add(stop_str)
        elif isinstance(stop_str, (list, tuple)):
            stop.update(s for s in stop_str if s)
        for tid in params.get("stop_token_ids") or []:
            if tid is not None:
                s = self.tokenizer.decode(tid)
                if s:
                    stop.add(s)
        greedy = temperature <= 1e-5
        ids = self.tokenizer(prompt).input_ids
        room = max(1, self.context_len - len(ids) - 1)
        req = Request(request_id=request_id, prompt=prompt, input_ids=list(ids), max_new_tokens=min(max_new_tokens, room),
                      do_sample=not greedy, temperature=None if greedy else temperature,
                      top_p=top_p if (not greedy and top_p < 1.0) else None, top_k=top_k if (not greedy and top_k > 0) else None)
        queue = await self.llm_engine.add_request(req)
        text, n, finish = "", 0, None
        try:
            while True:
                delta, finished = await queue.get()
                text += delta
                n += 1
                hit = min((text.find(s) for s in stop if s in text), default=-1)
                if hit >= 0:
                    text, finish, finished = text[:hit], "stop", True
                    self._abort(request_id)
                n_out = len(self.tokenizer(text, add_special_tokens=False).input_ids) if finished else n
                if finished and finish is None:
                    finish = "length" if n_out >= req.max_new_tokens else "stop"
                yield (json.dumps({"text": (prompt + text) if echo else text, "error_code": 0,
                                   "usage": {"prompt_tokens": len(ids), "completion_tokens": n_out,
                                             "total_tokens": len(ids) + n_out},
                                   "cumulative_logprob": [None], "finish_reason": finish if finished else None}) + "\0").encode()
                if finished:
                    break
        finally:
            if finish is None:                      # the consumer went away mid-stream
                self._abort(request_id)
            self.llm_engine.finish(request_id)

    async def generate(self, params):
        last = None
        async for last in self.generate_stream(params):
            pass
        return json.loads(last[:-1].decode())

    def count_token(self, params):
        return {"count": len(self.tokenizer(params["prompt"]).input_ids), "error_code": 0}

    def get_status(self):
        b = getattr(self.llm_engine, "batcher", None)
        return {"model_names": self.model_names, "speed": 1,
                "queue_length": 0 if b is None else len(b.waiting) + len(b.running)}


def create_app(worker: VLLMWorker):
    """FastChat's worker endpoints over ``worker`` (what fastchat/serve/vllm_worker.py registers on its module-level app)."""
    app = FastAPI()
    sem = asyncio.Semaphore(worker.limit_worker_concurrency)

    def background(request_id):
        tasks = BackgroundTasks()
        tasks.add_task(sem.release)
        tasks.add_task(worker._abort, request_id)       # no-op for a request that ran to its end
        return tasks

    @app.post("/worker_generate_stream")
    async def api_generate_stream(request: HttpRequest):
        params = await request.json()
        await sem.acquire()
        params["request_id"] = uuid.uuid4().hex
        return StreamingResponse(worker.generate_stream(params), background=background(params["request_id"]))

    @app.post("/worker_generate")
    async def api_generate(request: HttpRequest):
        params = await request.json()
        await sem.acquire()
        try:
            params["request_id"] = uuid.uuid4().hex
            return JSONResponse(await worker.generate(params))
        finally:
            sem.release()

    @app.post("/worker_get_status")
    async def api_get_status(request: HttpRequest):
        return worker.get_status()

    @app.post("/count_token")
    async def api_count_token(request: HttpRequest):
        return worker.count_token(await request.json())

    @app.post("/worker_get_conv_template")
    async def api_get_conv(request: HttpRequest):
        return {"conv": worker.conv_template}

    @app.post("/model_details")
    async def api_model_details(request: HttpRequest):
        return {"context_length": worker.context_len}

    return app


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", default="localhost")
    ap.add_argument("--port", type=int, default=21002)
    ap.add_argument("--worker-address", default="http://localhost:21002")
    ap.add_argument("--controller-address", default="http://localhost:21001")
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--model-names", type=lambda s: s.split(","))
    ap.add_argument("--load-in-low-bit", default="sym_int4")
    ap.add_argument("--limit-worker-concurrency", type=int, default=1024)
    ap.add_argument("--max-num-seqs", type=int, default=16)
    ap.add_argument("--max-model-len", type=int, default=4096)
    ap.add_argument("--no-register", action="store_true")
    ap.add_argument("--conv-template", default=None)
    a = ap.parse_args()
    import uvicorn
    w = VLLMWorker(a.controller_address, a.worker_address, worker_id, a.model_path, a.model_names, a.limit_worker_concurrency,
                   a.no_register, None, a.conv_template, a.load_in_low_bit, a.max_num_seqs, a.max_model_len)
    uvicorn.run(create_app(w), host=a.host, port=a.port, log_level="info")


if __name__ == "__main__":
    main()
