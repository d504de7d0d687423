"""TGI-style front end (bigdl_amd/serving/fastchat/tgi_api_server.py) against an in-process FastChat controller + worker
pair that speak the plain HTTP protocol of those components."""
import json

import httpx
import pytest
from fastapi import FastAPI, Request
from fastapi.responses import StreamingResponse
from fastapi.testclient import TestClient

from bigdl_amd.serving.fastchat.tgi_api_server import Settings, create_app


def _backend():
    app = FastAPI()
    seen = {}

    @app.post("/refresh_all_workers")
    async def refresh():
        return {}

    @app.post("/list_models")
    async def list_models():
        return {"models": ["tiny-llama"]}

    @app.post("/get_worker_address")
    async def addr(req: Request):
        body = await req.json()
        return {"address": "http://worker" if body["model"] == "tiny-llama" else ""}

    @app.post("/model_details")
    async def details():
        return {"context_length": 64}

    @app.post("/count_token")
    async def count(req: Request):
        return {"count": len((await req.json())["prompt"].split())}

    @app.post("/worker_get_conv_template")
    async def conv():
        return {"conv": {"roles": ["USER", "ASSISTANT"], "sep": "\n", "system_message": "", "stop_str": "</s>", "stop_token_ids": [2]}}

    @app.post("/worker_generate_stream")
    async def stream(req: Request):
        body = await req.json()
        seen["params"] = body

        def chunks():
            text = ""
            words = ["Hello", " from", " the", " worker"]
            for i, w in enumerate(words):
                text += w
                last = i == len(words) - 1
                yield json.dumps({"text": text, "error_code": 0, "finish_reason": "stop" if last else None,
                                  "usage": {"prompt_tokens": 3, "completion_tokens": i + 1, "total_tokens": 4 + i}}).encode() + b"\0"
        return StreamingResponse(chunks())
    return app, seen


@pytest.fixture
def client():
    backend, seen = _backend()
    s = Settings()
    s.controller_address = "http://controller"
    app = create_app(s, client_factory=lambda: httpx.AsyncClient(transport=httpx.ASGITransport(app=backend), base_url="http://x"))
    return TestClient(app), seen, s


def test_generate_and_models(client):
    c, seen, _ = client
    assert [m["id"] for m in c.get("/v1/models").json()["data"]] == ["tiny-llama"]
    r = c.post("/generate", json={"inputs": "say hello please", "parameters": {"max_new_tokens": 500, "temperature": 0.7, "top_k": 5}})
    assert r.status_code == 200, r.text
    body = r.json()
    assert body["generated_text"] == "Hello from the worker"
    assert body["details"]["best_of_sequences"][0]["finish_reason"] == "stop"
    assert body["usage"] == {"prompt_tokens": 3, "total_tokens": 7, "completion_tokens": 4}
    p = seen["params"]
    assert p["prompt"] == "say hello please" and p["max_new_tokens"] == 61          # clamped to the 64-token window
    assert p["temperature"] == 0.7 and p["top_k"] == 5 and p["model"] == "tiny-llama"


def test_stream_events_and_validation(client):
    c, seen, s = client
    with c.stream("POST", "/generate_stream", json={"inputs": "hi", "parameters": {"max_new_tokens": 8}}) as r:
        lines = [ln for ln in r.iter_lines() if ln.startswith("data: ")]
    assert lines[-1] == "data: [DONE]"
    events = [json.loads(ln[6:]) for ln in lines[:-1]]
    assert "".join(e["token"]["text"] for e in events) == "Hello from the worker"
    assert events[-1]["generated_text"] == "Hello from the worker" and events[-1]["details"]["finish_reason"] == "stop"
    assert seen["params"]["temperature"] == 0.0                                      # greedy unless sampling is asked for
    bad = c.post("/generate", json={"inputs": "x", "parameters": {"temperature": 3.0}})
    assert bad.status_code == 400 and bad.json()["code"] == 40302
    unknown = c.post("/generate", json={"inputs": "x", "model": "other", "parameters": {}})
    assert unknown.status_code == 400 and unknown.json()["code"] == 40301
    s.apply_template = True
    c.post("/generate", json={"inputs": "hi", "parameters": {"max_new_tokens": 4}})
    assert seen["params"]["prompt"] == "USER: hi\nASSISTANT:" and seen["params"]["stop"] == ["</s>"]
    s.apply_template = False
    s.api_keys = ["k1"]
    assert c.post("/generate", json={"inputs": "x", "parameters": {}}).status_code == 401
    ok = c.post("/generate", json={"inputs": "x", "parameters": {}}, headers={"Authorization": "Bearer k1"})
    assert ok.status_code == 200
