"""TGI-style REST front end for FastChat deployments: ``POST /generate`` and ``POST /generate_stream`` in the request /
response shapes of Hugging Face text-generation-inference, served from the workers registered with a FastChat
controller.

Capability: python/llm/src/ipex_llm/serving/fastchat/tgi_api_server.py (routes :419-547, worker lookup :390-417,
generation parameters :291-388, the streaming generator :548-631, API-key check :135-155) with the wire schema of
tgi_api_protocol.py:70-135.

This is a controller *client*: it speaks the controller's and the workers' plain HTTP / JSON protocol
(``/get_worker_address``, ``/list_models``, ``/worker_get_conv_template``, ``/count_token``, ``/model_details``,
``/worker_generate_stream`` with NUL-delimited JSON chunks), so it needs neither the ``fastchat`` package nor a GPU and
works with this package's ``BigDLLLMWorker`` or any other FastChat worker.  Prompts are passed through as given
(TGI's ``inputs`` is already a flat prompt; the reference builds a one-turn conversation from the worker's template,
which is what ``--apply-template`` switches on here).

    python -m bigdl_amd.serving.fastchat.tgi_api_server --controller-address http://localhost:21001 --port 8000
"""
from __future__ import annotations

import argparse
import json
import os
import time
from typing import Any, AsyncIterator, Dict, List, Optional

import httpx
from fastapi import Depends, FastAPI, HTTPException
from fastapi.responses import JSONResponse, StreamingResponse
from fastapi.security.http import HTTPAuthorizationCredentials, HTTPBearer
from pydantic import BaseModel, Field

WORKER_API_TIMEOUT = int(os.environ.get("FASTCHAT_WORKER_API_TIMEOUT", 100))
PSEUDO_INFINITE = 1 << 31


# ---------------------------------------------------------------------------------------------
# wire schema
# ---------------------------------------------------------------------------------------------
class Grammar(BaseModel):
    type: str = "json"
    value: str = "string"


class GenerateParameters(BaseModel):
    best_of: Optional[int] = 1
    decoder_input_details: Optional[bool] = True
    details: Optional[bool] = False
    do_sample: Optional[bool] = False
    frequency_penalty: Optional[float] = 0.1
    grammar: Optional[Grammar] = None
    max_new_tokens: Optional[int] = None
    repetition_penalty: Optional[float] = 1.0
    return_full_text: Optional[bool] = False
    seed: Optional[float] = None
    stop: Optional[List[str]] = Field(default_factory=list)
    temperature: Optional[float] = 1.0
    top_k: Optional[int] = PSEUDO_INFINITE
    top_n_tokens: Optional[int] = 5
    top_p: Optional[float] = 1.0
    truncate: Optional[bool] = False
    typical_p: Optional[float] = 0.95
    watermark: Optional[bool] = True


class GenerateRequest(BaseModel):
    inputs: str
    parameters: GenerateParameters = Field(default_factory=GenerateParameters)
    model: Optional[str] = ""


class UsageInfo(BaseModel):
    prompt_tokens: int = 0
    total_tokens: int = 0
    completion_tokens: Optional[int] = 0


class Sequence(BaseModel):
    index: int
    message: Dict[str, str]
    finish_reason: Optional[str] = None
    generated_text: str
    generated_tokens: Optional[int] = 0


class Details(BaseModel):
    best_of_sequences: List[Sequence]


class GenerateResponse(BaseModel):
    details: Details
    generated_text: str
    usage: UsageInfo


class StreamToken(BaseModel):
    id: int
    text: str
    logprob: Optional[float] = None
    special: bool = False


class StreamDetails(BaseModel):
    finish_reason: Optional[str] = None
    generated_tokens: int = 0
    seed: Optional[float] = None


class StreamResponse(BaseModel):
    token: StreamToken
    generated_text: Optional[str] = None
    details: Optional[StreamDetails] = None


# ---------------------------------------------------------------------------------------------
# server
# ---------------------------------------------------------------------------------------------
class Settings:
    controller_address: str = "http://localhost:21001"
    api_keys: Optional[List[str]] = None
    apply_template: bool = False


def create_app(settings: Optional[Settings] = None, client_factory=None) -> FastAPI:
    """``client_factory() -> httpx.AsyncClient``: how the controller and the workers are reached (tests plug an
    in-process transport in here)."""
    settings = settings or Settings()
    client_factory = client_factory or (lambda: httpx.AsyncClient(timeout=httpx.Timeout(WORKER_API_TIMEOUT)))
    app = FastAPI(title="TGI-style API over FastChat workers")
    bearer = HTTPBearer(auto_error=False)

    async def check_api_key(auth: Optional[HTTPAuthorizationCredentials] = Depends(bearer)):
        if settings.api_keys:
            if auth is None or auth.credentials not in settings.api_keys:
                raise HTTPException(status_code=401, detail={"error": {"message": "invalid API key", "type": "invalid_request_error",
                                                                         "code": "invalid_api_key"}})
        return None

    def error(code: int, message: str) -> JSONResponse:
        return JSONResponse({"object": "error", "message": message, "code": code}, status_code=400)

    async def post_json(url: str, payload: Optional[dict] = None) -> Any:
        async with client_factory() as c:
            r = await c.post(url, json=payload or {})
            r.raise_for_status()
            return r.json()

    async def models() -> List[str]:
        await post_json(settings.controller_address + "/refresh_all_workers")
        return sorted((await post_json(settings.controller_address + "/list_models"))["models"])

    async def worker_address(model: str) -> str:
        addr = (await post_json(settings.controller_address + "/get_worker_address", {"model": model})).get("address", "")
        if not addr:
            raise HTTPException(status_code=400, detail=f"no available worker for {model}")
        return addr

    async def resolve(req: GenerateRequest):
        """-> (model name, worker address) or an error response"""
        names = await models()
        model = req.model or (names[-1] if names else "")
        if model not in names:
            return None, error(40301, f"Only {'&&'.join(names)} allowed now, your model {model}")
        return (model, await worker_address(model)), None

    def validate(p: GenerateParameters) -> Optional[JSONResponse]:
        checks = [(p.max_new_tokens is not None and p.max_new_tokens <= 0, f"{p.max_new_tokens} is less than the minimum of 1 - 'max_new_tokens'"),
                  (p.best_of is not None and p.best_of <= 0, f"{p.best_of} is less than the minimum of 1 - 'best_of'"),
                  (p.temperature is not None and not 0 <= p.temperature <= 2, f"{p.temperature} is out of [0, 2] - 'temperature'"),
                  (p.top_p is not None and not 0 <= p.top_p <= 1, f"{p.top_p} is out of [0, 1] - 'top_p'"),
                  (p.top_k is not None and p.top_k < 1 and p.top_k != -1, f"{p.top_k} is out of range - 'top_k'")]
        for bad, msg in checks:
            if bad:
                return error(40302, msg)
        return None

    async def gen_params(model: str, addr: str, req: GenerateRequest) -> Dict[str, Any]:
        p = req.parameters
        prompt, stops, stop_ids = req.inputs, list(p.stop or []), []
        if settings.apply_template:
            conv = (await post_json(addr + "/worker_get_conv_template", {"model": model}))["conv"]
            roles, sep = conv.get("roles", ["USER", "ASSISTANT"]), conv.get("sep") or "\n"
            prompt = f"{conv.get('system_message', '')}{sep if conv.get('system_message') else ''}{roles[0]}: {req.inputs}{sep}{roles[1]}:"
            if conv.get("stop_str"):
                stops += [conv["stop_str"]] if isinstance(conv["stop_str"], str) else list(conv["stop_str"])
            stop_ids = conv.get("stop_token_ids") or []
        temperature = p.temperature if p.do_sample or (p.temperature not in (None, 1.0)) else 0.0
        out = {"model": model, "prompt": prompt, "temperature": temperature, "top_p": p.top_p,
               "top_k": -1 if p.top_k in (None, PSEUDO_INFINITE) else p.top_k, "presence_penalty": 0.0,
               "frequency_penalty": p.frequency_penalty, "repetition_penalty": p.repetition_penalty,
               "max_new_tokens": p.max_new_tokens, "echo": bool(p.return_full_text), "stop": sorted(set(stops)),
               "stop_token_ids": stop_ids, "logprobs": None, "best_of": p.best_of}
        # clamp the budget to the worker's context window
        ctx = (await post_json(addr + "/model_details", {"model": model}))["context_length"]
        n_in = (await post_json(addr + "/count_token", {"model": model, "prompt": prompt}))["count"]
        room = ctx - n_in
        if room <= 0:
            raise HTTPException(status_code=400, detail=f"This model's maximum context length is {ctx} tokens, the prompt has {n_in}")
        out["max_new_tokens"] = min(out["max_new_tokens"] or room, room)
        return out

    async def worker_stream(addr: str, payload: Dict[str, Any]) -> AsyncIterator[Dict[str, Any]]:
        """chunks of the worker's NUL-delimited JSON stream"""
        async with client_factory() as c:
            async with c.stream("POST", addr + "/worker_generate_stream", json=payload) as r:
                buf = b""
                async for raw in r.aiter_raw():
                    buf += raw
                    while True:
                        i = buf.find(b"\0")
                        if i < 0:
                            break
                        piece, buf = buf[:i], buf[i + 1:]
                        if piece:
                            yield json.loads(piece.decode())

    @app.get("/v1/models", dependencies=[Depends(check_api_key)])
    async def list_models():
        now = int(time.time())
        return {"object": "list", "data": [{"id": m, "object": "model", "created": now, "owned_by": "fastchat", "root": m}
                                           for m in await models()]}

    @app.get("/health")
    async def health():
        return {"status": "ok"}

    @app.post("/generate", dependencies=[Depends(check_api_key)])
    async def generate(req: GenerateRequest):
        bad = validate(req.parameters)
        if bad is not None:
            return bad
        found, err = await resolve(req)
        if err is not None:
            return err
        model, addr = found
        params = await gen_params(model, addr, req)
        seqs, usage = [], UsageInfo()
        for i in range(max(1, req.parameters.best_of or 1)):
            last = None
            async for chunk in worker_stream(addr, params):
                last = chunk
            if last is None:
                return error(50002, "the worker returned no data")
            if last.get("error_code", 0) != 0:
                return error(last["error_code"], last.get("text", "worker error"))
            u = last.get("usage") or {}
            seqs.append(Sequence(index=i, message={"role": "assistant", "content": last["text"]},
                                 finish_reason=last.get("finish_reason") or "stop", generated_text=last["text"],
                                 generated_tokens=u.get("completion_tokens", 0)))
            usage.prompt_tokens += u.get("prompt_tokens", 0)
            usage.completion_tokens += u.get("completion_tokens", 0)
            usage.total_tokens += u.get("total_tokens", 0)
        return GenerateResponse(details=Details(best_of_sequences=seqs), generated_text=seqs[0].generated_text, usage=usage)

    @app.post("/generate_stream", dependencies=[Depends(check_api_key)])
    async def generate_stream(req: GenerateRequest):
        bad = validate(req.parameters)
        if bad is not None:
            return bad
        found, err = await resolve(req)
        if err is not None:
            return err
        model, addr = found
        params = await gen_params(model, addr, req)

        async def events():
            sent, n = "", 0
            async for chunk in worker_stream(addr, params):
                if chunk.get("error_code", 0) != 0:
                    yield f"data: {json.dumps(chunk, ensure_ascii=False)}\n\n"
                    break
                text = chunk.get("text", "")
                delta, sent = text[len(sent):] if text.startswith(sent) else text, text
                finish = chunk.get("finish_reason")
                if not delta and not finish:
                    continue
                ev = StreamResponse(token=StreamToken(id=n, text=delta))
                n += 1
                if finish:
                    u = chunk.get("usage") or {}
                    ev.generated_text = text
                    ev.details = StreamDetails(finish_reason=finish, generated_tokens=u.get("completion_tokens", n),
                                               seed=req.parameters.seed)
                yield f"data: {ev.model_dump_json(exclude_none=True)}\n\n"
            yield "data: [DONE]\n\n"
        return StreamingResponse(events(), media_type="text/event-stream")

    return app


def main(argv=None):
    ap = argparse.ArgumentParser(description="TGI-style RESTful API server over FastChat workers")
    ap.add_argument("--host", default="localhost")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--controller-address", default="http://localhost:21001")
    ap.add_argument("--api-keys", type=lambda s: s.split(","), default=None, help="comma separated list of accepted API keys")
    ap.add_argument("--apply-template", action="store_true", help="wrap `inputs` in the worker's conversation template")
    a = ap.parse_args(argv)
    s = Settings()
    s.controller_address, s.api_keys, s.apply_template = a.controller_address, a.api_keys, a.apply_template
    import uvicorn
    uvicorn.run(create_app(s), host=a.host, port=a.port, log_level="info")


if __name__ == "__main__":
    main()
