"""Lightweight HTTP serving: TGI-style ``/generate`` + ``/generate_stream`` and OpenAI-style ``/v1/chat/completions``,
``/v1/completions``, ``/v1/models``; ``/health`` and Prometheus ``/metrics``.

Reference: python/llm/src/ipex_llm/serving/fastapi/api_server.py:103-437 (``FastApp``, the SSE generators, prompt
assembly from chat messages, background ``process_requests`` loop started at app startup).
"""
from __future__ import annotations

import asyncio
import os
import json
import uuid
import logging
from typing import List, Optional

from fastapi import FastAPI, HTTPException
from fastapi import Request as HttpRequest
from fastapi.responses import JSONResponse, PlainTextResponse, StreamingResponse

from .model_worker import ModelWorker, Request
from .openai_protocol import (ChatCompletionRequest, ChatCompletionResponse, ChatCompletionResponseChoice,
                              ChatCompletionResponseStreamChoice, ChatCompletionStreamResponse, ChatMessage,
                              CompletionRequest, CompletionResponse, CompletionResponseChoice, CompletionStreamResponse,
                              DeltaMessage, ModelCard, ModelList, UsageInfo)
from .tgi_protocol import GenerateResponse, InputsRequest, Parameters

logger = logging.getLogger(__name__)


def get_prompt(messages, tokenizer=None) -> str:
    """Chat messages -> prompt text: the tokenizer's chat template when it has one, a plain role-tagged transcript
    otherwise."""
    msgs = []
    for m in messages:
        content = m.content
        if isinstance(content, list):             # OpenAI multi-part content: keep the text parts
            content = "".join(p.get("text", "") for p in content if p.get("type", "text") == "text")
        msgs.append({"role": m.role, "content": content})
    if tokenizer is not None and getattr(tokenizer, "chat_template", None):
        return tokenizer.apply_chat_template(msgs, tokenize=False, add_generation_prompt=True)
    out = ""
    for m in msgs:
        if m["role"] == "system":
            out += f"<<SYS>>\n{m['content']}\n<</SYS>>\n\n"
        elif m["role"] == "user":
            out += f"[INST] {m['content']} [/INST] "
        elif m["role"] == "assistant":
            out += f"{m['content']} "
        else:
            raise HTTPException(status_code=400, detail=f"Unknown role: {m['role']}")
    return out.strip()


def parse_multipart(content_type: str, body: bytes) -> dict:
    """multipart/form-data -> {field: bytes or str} with the standard library's MIME parser (the optional
    ``python-multipart`` dependency of FastAPI's ``UploadFile`` is not required)."""
    from email.parser import BytesParser
    from email.policy import HTTP
    msg = BytesParser(policy=HTTP).parsebytes(b"Content-Type: " + content_type.encode() + b"\r\n\r\n" + body)
    out = {}
    for part in msg.iter_parts():
        name = part.get_param("name", header="content-disposition")
        data = part.get_payload(decode=True)
        if part.get_filename() is None:
            data = data.decode("utf-8", "replace").strip()
        out[name] = data
    return out


def decode_wav(data: bytes, target_rate: int = 16000):
    """PCM WAV bytes -> mono float32 numpy array at ``target_rate`` (linear resampling)."""
    import io
    import wave

    import numpy as np
    with wave.open(io.BytesIO(data)) as w:
        rate, width, ch = w.getframerate(), w.getsampwidth(), w.getnchannels()
        raw = w.readframes(w.getnframes())
    dt = {1: np.uint8, 2: np.int16, 4: np.int32}[width]
    x = np.frombuffer(raw, dtype=dt).astype(np.float32)
    x = (x - 128.0) / 128.0 if width == 1 else x / float(2 ** (8 * width - 1))
    if ch > 1:
        x = x.reshape(-1, ch).mean(1)
    if rate != target_rate and x.size:
        n = int(round(x.size * target_rate / rate))
        x = np.interp(np.linspace(0, x.size - 1, n), np.arange(x.size), x).astype(np.float32)
    return x


def _stops(stop) -> List[str]:
    if stop is None:
        return []
    return [stop] if isinstance(stop, str) else list(stop)


class FastApp:
    def __init__(self, model=None, mytokenizer=None, myprocessor=None, worker: Optional[ModelWorker] = None,
                 served_model_name: str = "default", max_new_tokens_default: int = 128):
        self.worker = worker or ModelWorker(model=model, tokenizer=mytokenizer)
        self.tokenizer = self.worker.tokenizer
        self.processor = myprocessor
        self.name = served_model_name
        self.default_max = max_new_tokens_default
        self.token_timeout = float(os.environ.get("IPEX_LLM_FASTAPI_TIMEOUT", 60))
        self.max_consecutive_errors = int(os.environ.get("BIGDL_SERVING_MAX_ERRORS", 8))
        self.loop_dead = False
        self.app = FastAPI(title="bigdl_amd lightweight serving")
        self._task = None
        self._routes()

    # ---------------------------------------------------------------------------------------
    async def _loop(self):
        """Drives the worker.  A failed step is logged with its traceback and the loop goes on (the worker has already
        failed the requests the step carried); ``max_consecutive_errors`` failures in a row mean the worker cannot make
        progress at all (device fault, corrupted state): the loop stops and ``/health`` reports it instead of spinning."""
        errors = 0
        while True:
            try:
                await self.worker.process_step()
                errors = 0
            except asyncio.CancelledError:
                break
            except Exception:
                errors += 1
                logger.exception("serving step failed (%d in a row)", errors)
                if errors >= self.max_consecutive_errors:
                    self.loop_dead = True
                    logger.error("serving loop stopped after %d consecutive failures", errors)
                    break
                await asyncio.sleep(min(0.05 * errors, 1.0))

    def _request(self, prompt, p: Parameters) -> Request:
        return Request(request_id=uuid.uuid4().hex, prompt=prompt, max_new_tokens=p.max_new_tokens or self.default_max,
                       min_new_tokens=p.min_new_tokens or 0, do_sample=bool(p.do_sample), temperature=p.temperature,
                       top_p=p.top_p, top_k=p.top_k, repetition_penalty=p.repetition_penalty, stop=p.stop or [])

    async def _deltas(self, req: Request):
        """Async iterator of text deltas with stop-string handling."""
        q = await self.worker.add_request(req)
        text = ""
        try:
            while True:
                try:
                    # IPEX_LLM_FASTAPI_TIMEOUT (seconds, default 60): a request whose worker stops producing tokens
                    # ends with what it has instead of holding the connection (reference api_server.py:114-116)
                    delta, done = await asyncio.wait_for(q.get(), timeout=self.token_timeout)
                except asyncio.TimeoutError:
                    yield "", True
                    return
                text += delta
                cut = min((text.find(s) for s in req.stop if s and s in text), default=-1)
                if cut >= 0:
                    keep = cut - (len(text) - len(delta))
                    if keep > 0:
                        yield delta[:keep], False
                    yield "", True
                    return
                yield delta, done
                if done:
                    return
        finally:
            self.worker.finish(req.request_id)

    async def _collect(self, req: Request):
        out = ""
        async for delta, _ in self._deltas(req):
            out += delta
        return out

    def _usage(self, prompt: str, completion: str) -> UsageInfo:
        n_p = len(self.tokenizer(prompt).input_ids)
        n_c = len(self.tokenizer(completion, add_special_tokens=False).input_ids) if completion else 0
        return UsageInfo(prompt_tokens=n_p, completion_tokens=n_c, total_tokens=n_p + n_c)

    @staticmethod
    def _params(r) -> Parameters:
        return Parameters(max_new_tokens=r.max_tokens or 128, min_new_tokens=r.min_tokens, temperature=r.temperature,
                          top_p=r.top_p, top_k=r.top_k, repetition_penalty=r.repetition_penalty, stop=_stops(r.stop),
                          do_sample=bool(r.temperature and r.temperature > 0 and r.temperature != 1.0))

    # ---------------------------------------------------------------------------------------
    def _routes(self):
        app = self.app

        @app.on_event("startup")
        async def _startup():
            self._task = asyncio.create_task(self._loop())

        @app.on_event("shutdown")
        async def _shutdown():
            if self._task is not None:
                self._task.cancel()

        @app.get("/health")
        async def health():
            if self.loop_dead:
                return JSONResponse({"status": "failed", "detail": "serving loop stopped after repeated errors"}, status_code=503)
            return JSONResponse({"status": "ok", "waiting": self.worker.waiting_requests.qsize()})

        @app.get("/v1/models")
        async def models():
            return ModelList(data=[ModelCard(id=self.name)])

        @app.get("/metrics")
        async def metrics():
            s = self.worker.stats
            lines = ["# TYPE bigdl_requests_total counter", f"bigdl_requests_total {s['requests']}",
                     "# TYPE bigdl_prompt_tokens_total counter", f"bigdl_prompt_tokens_total {s['prompt_tokens']}",
                     "# TYPE bigdl_generated_tokens_total counter", f"bigdl_generated_tokens_total {s['generated_tokens']}",
                     "# TYPE bigdl_busy_seconds_total counter", f"bigdl_busy_seconds_total {s['busy_s']:.6f}",
                     "# TYPE bigdl_waiting_requests gauge", f"bigdl_waiting_requests {self.worker.waiting_requests.qsize()}"]
            if s["first_token_s"]:
                lines += ["# TYPE bigdl_first_token_seconds gauge",
                          f"bigdl_first_token_seconds {sum(s['first_token_s']) / len(s['first_token_s']):.6f}"]
            if s["per_token_s"]:
                lines += ["# TYPE bigdl_next_token_seconds gauge",
                          f"bigdl_next_token_seconds {sum(s['per_token_s']) / len(s['per_token_s']):.6f}"]
            return PlainTextResponse("\n".join(lines) + "\n")

        @app.post("/generate")
        async def generate(inputs_request: InputsRequest):
            if inputs_request.stream:
                return await generate_stream(inputs_request)
            req = self._request(inputs_request.inputs, inputs_request.parameters or Parameters())
            text = await self._collect(req)
            n = len(self.tokenizer(text, add_special_tokens=False).input_ids) if text else 0
            return GenerateResponse(generated_text=text, generate_tokens=n,
                                    finish_reason="length" if n >= req.max_new_tokens else "stop")

        @app.post("/generate_stream")
        async def generate_stream(inputs_request: InputsRequest):
            req = self._request(inputs_request.inputs, inputs_request.parameters or Parameters())

            async def gen():
                full = ""
                async for delta, done in self._deltas(req):
                    full += delta
                    payload = {"token": {"text": delta}, "generated_text": full if done else None}
                    yield f"data: {json.dumps(payload, ensure_ascii=False)}\n\n"
                yield "data: [DONE]\n\n"
            return StreamingResponse(gen(), media_type="text/event-stream")

        @app.post("/v1/chat/completions")
        async def chat_completions(request: ChatCompletionRequest):
            prompt = get_prompt(request.messages, self.tokenizer)
            req = self._request(prompt, self._params(request))
            if request.stream:
                rid = f"chatcmpl-{req.request_id}"

                async def gen():
                    first = ChatCompletionStreamResponse(id=rid, model=request.model, choices=[
                        ChatCompletionResponseStreamChoice(index=0, delta=DeltaMessage(role="assistant"))])
                    yield f"data: {first.model_dump_json(exclude_unset=True)}\n\n"
                    full = ""
                    async for delta, done in self._deltas(req):
                        full += delta
                        if delta:
                            chunk = ChatCompletionStreamResponse(id=rid, model=request.model, choices=[
                                ChatCompletionResponseStreamChoice(index=0, delta=DeltaMessage(content=delta))])
                            yield f"data: {chunk.model_dump_json(exclude_unset=True)}\n\n"
                    last = ChatCompletionStreamResponse(id=rid, model=request.model, choices=[
                        ChatCompletionResponseStreamChoice(index=0, delta=DeltaMessage(), finish_reason="stop")],
                        usage=self._usage(prompt, full))
                    yield f"data: {last.model_dump_json(exclude_unset=True)}\n\n"
                    yield "data: [DONE]\n\n"
                return StreamingResponse(gen(), media_type="text/event-stream")
            text = await self._collect(req)
            return ChatCompletionResponse(model=request.model, usage=self._usage(prompt, text), choices=[
                ChatCompletionResponseChoice(index=0, message=ChatMessage(role="assistant", content=text),
                                             finish_reason="stop")])

        @app.post("/v1/audio/transcriptions")
        async def transcriptions(request: HttpRequest):
            """OpenAI audio API: multipart form with ``file`` (PCM WAV) and optional ``language`` / ``response_format``
            (reference api_server.py:403-426)."""
            ctype = request.headers.get("content-type", "")
            body = await request.body()
            if "multipart/form-data" in ctype:
                form = parse_multipart(ctype, body)
            else:
                form = {"file": body}
            if not isinstance(form.get("file"), (bytes, bytearray)) or not form["file"]:
                raise HTTPException(status_code=400, detail="missing audio `file`")
            try:
                audio = decode_wav(bytes(form["file"]))
            except Exception as e:
                raise HTTPException(status_code=400, detail=f"unsupported audio (PCM WAV expected): {e}")
            loop = asyncio.get_running_loop()
            text = await loop.run_in_executor(None, self.transcribe, audio, form.get("language"))
            if form.get("response_format") == "text":
                return PlainTextResponse(text)
            return JSONResponse({"text": text})

        @app.post("/v1/completions")
        async def completions(request: CompletionRequest):
            prompt = request.prompt
            if isinstance(prompt, list):
                if prompt and isinstance(prompt[0], int):
                    prompt = self.tokenizer.decode(prompt)
                elif len(prompt) == 1 and isinstance(prompt[0], str):
                    prompt = prompt[0]
                elif len(prompt) == 1:
                    prompt = self.tokenizer.decode(prompt[0])
                else:
                    raise HTTPException(status_code=400, detail="one prompt per request")
            req = self._request(prompt, self._params(request))
            if request.stream:
                rid = f"cmpl-{req.request_id}"

                async def gen():
                    full = ""
                    async for delta, done in self._deltas(req):
                        full += delta
                        chunk = CompletionStreamResponse(id=rid, model=request.model, choices=[
                            CompletionResponseChoice(index=0, text=delta, finish_reason="stop" if done else None)],
                            usage=self._usage(prompt, full) if done else None)
                        yield f"data: {chunk.model_dump_json(exclude_unset=True)}\n\n"
                    yield "data: [DONE]\n\n"
                return StreamingResponse(gen(), media_type="text/event-stream")
            text = await self._collect(req)
            return CompletionResponse(model=request.model, usage=self._usage(prompt, text), choices=[
                CompletionResponseChoice(index=0, text=(prompt + text) if request.echo else text, finish_reason="stop")])


    def transcribe(self, audio, language: Optional[str] = None, max_new_tokens: int = 256) -> str:
        """Speech -> text with a Whisper-style seq2seq model (``processor`` = feature extractor + tokenizer)."""
        import torch
        if self.processor is None:
            raise HTTPException(status_code=400, detail="this server was started without an audio processor")
        model = self.worker.model
        feats = self.processor(audio, sampling_rate=16000, return_tensors="pt").input_features
        p = next(p for p in model.parameters() if p.numel() > 0)
        dtype = p.dtype if p.dtype in (torch.float16, torch.bfloat16, torch.float32) else torch.float16
        kw = {}
        if language and hasattr(self.processor, "get_decoder_prompt_ids"):
            kw["forced_decoder_ids"] = self.processor.get_decoder_prompt_ids(language=language, task="transcribe")
        limit = getattr(model.config, "max_target_positions", None)
        if limit:
            max_new_tokens = max(1, min(max_new_tokens, limit - 4))      # room for the start / language / task tokens
        with torch.inference_mode():
            ids = model.generate(feats.to(p.device, dtype), max_new_tokens=max_new_tokens, **kw)
        return self.processor.batch_decode(ids, skip_special_tokens=True)[0]


def serve(repo_id_or_model_path: str, low_bit: str = "sym_int4", port: int = 8000, host: str = "0.0.0.0",
          served_model_name: Optional[str] = None, model_type: str = "normal", max_num_seqs: int = 0,
          max_seq_len: int = 4096, max_prefilled_seqs: int = 8):
    """Load a model and serve it (reference example: python/llm/example/GPU/Lightweight-Serving/lightweight_serving.py).
    ``max_num_seqs`` > 0: iteration-level batching over that many cache slots of the native engine
    (``BatchedModelWorker``) instead of one request at a time."""
    import uvicorn
    if max_num_seqs > 0:
        from .model_worker import BatchedModelWorker
        worker = BatchedModelWorker(repo_id_or_model_path, low_bit, model_type, max_num_seqs=max_num_seqs,
                                    max_seq_len=max_seq_len, max_prefilled_seqs=max_prefilled_seqs)
    else:
        worker = ModelWorker(repo_id_or_model_path, low_bit, model_type)
    fast = FastApp(worker=worker, served_model_name=served_model_name or repo_id_or_model_path)
    uvicorn.run(fast.app, host=host, port=port)


def main():
    import argparse
    ap = argparse.ArgumentParser(description="bigdl_amd lightweight serving")
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--served-model-name", default=None)
    ap.add_argument("--model-type", default="normal")
    ap.add_argument("--max-num-seqs", type=int, default=0, help="> 0: continuous batching over this many cache slots")
    ap.add_argument("--max-seq-len", type=int, default=4096)
    ap.add_argument("--max-prefilled-seqs", type=int, default=8)
    a = ap.parse_args()
    serve(a.repo_id_or_model_path, a.low_bit, a.port, a.host, a.served_model_name, a.model_type, a.max_num_seqs,
          a.max_seq_len, a.max_prefilled_seqs)


if __name__ == "__main__":
    main()
