"""Lightweight serving routes against a tiny low-bit model (FastAPI TestClient, CPU)."""
import json

import pytest
import torch
from fastapi.testclient import TestClient
from tokenizers import Tokenizer
from tokenizers.models import WordLevel
from tokenizers.pre_tokenizers import Whitespace
from transformers import LlamaConfig, LlamaForCausalLM, PreTrainedTokenizerFast


@pytest.fixture(scope="module")
def client():
    from bigdl_amd.optimize import optimize_model
    from bigdl_amd.serving.fastapi import FastApp
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=256, tie_word_embeddings=False)
    model = LlamaForCausalLM(cfg).eval()
    model.generation_config.eos_token_id = None
    model = optimize_model(model, low_bit="sym_int8")
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    vocab.update({f"w{i}": i + 3 for i in range(317)})
    t = Tokenizer(WordLevel(vocab, unk_token="<unk>"))
    t.pre_tokenizer = Whitespace()
    tok = PreTrainedTokenizerFast(tokenizer_object=t, bos_token="<s>", eos_token="</s>", unk_token="<unk>")
    fast = FastApp(model, tok, served_model_name="tiny")
    with TestClient(fast.app) as c:
        c.model, c.tok = model, tok
        yield c


def _greedy_text(client, prompt, n):
    ids = client.tok(prompt, return_tensors="pt").input_ids
    out = client.model.generate(ids, max_new_tokens=n, do_sample=False)
    return client.tok.decode(out[0, ids.shape[1]:], skip_special_tokens=True)


def test_generate_and_stream_match_direct_generation(client):
    ref = _greedy_text(client, "w1 w2 w3", 6)
    r = client.post("/generate", json={"inputs": "w1 w2 w3", "parameters": {"max_new_tokens": 6}})
    assert r.status_code == 200 and r.json()["generated_text"] == ref and r.json()["generate_tokens"] == 6
    with client.stream("POST", "/generate_stream", json={"inputs": "w1 w2 w3", "parameters": {"max_new_tokens": 6}}) as s:
        chunks = [l[len("data: "):] for l in s.iter_lines() if l.startswith("data: ")]
    assert chunks[-1] == "[DONE]"
    text = "".join(json.loads(c)["token"]["text"] for c in chunks[:-1])
    assert text == ref


def test_openai_routes(client):
    ref = _greedy_text(client, "w5 w6", 5)
    r = client.post("/v1/completions", json={"model": "tiny", "prompt": "w5 w6", "max_tokens": 5})
    body = r.json()
    assert r.status_code == 200 and body["choices"][0]["text"] == ref
    assert body["usage"]["prompt_tokens"] == 2 and body["usage"]["completion_tokens"] == 5
    r = client.post("/v1/chat/completions", json={"model": "tiny", "max_tokens": 4,
                                                  "messages": [{"role": "user", "content": "w7 w8"}]})
    assert r.status_code == 200 and r.json()["choices"][0]["message"]["role"] == "assistant"
    with client.stream("POST", "/v1/chat/completions", json={"model": "tiny", "max_tokens": 4, "stream": True,
                                                             "messages": [{"role": "user", "content": "w7 w8"}]}) as s:
        lines = [l[len("data: "):] for l in s.iter_lines() if l.startswith("data: ")]
    assert lines[-1] == "[DONE]" and json.loads(lines[0])["choices"][0]["delta"]["role"] == "assistant"
    streamed = "".join(json.loads(l)["choices"][0]["delta"].get("content") or "" for l in lines[1:-1])
    assert streamed == r.json()["choices"][0]["message"]["content"]
    assert client.get("/v1/models").json()["data"][0]["id"] == "tiny"


def test_stop_strings_health_and_metrics(client):
    ref = _greedy_text(client, "w9 w10", 8)
    words = ref.split()
    stop = words[2]
    r = client.post("/v1/completions", json={"prompt": "w9 w10", "max_tokens": 8, "stop": [stop]})
    got = r.json()["choices"][0]["text"]
    assert stop not in got and ref.startswith(got)
    assert client.get("/health").json()["status"] == "ok"
    m = client.get("/metrics").text
    assert "bigdl_requests_total" in m and "bigdl_generated_tokens_total" in m
