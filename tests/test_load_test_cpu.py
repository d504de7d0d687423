"""Serving load-test client: statistics and SSE parsing, and one level against an in-process fake stream."""
import importlib.util
import os
import time

spec = importlib.util.spec_from_file_location(
    "serving_load_test", os.path.join(os.path.dirname(__file__), "..", "benchmark", "serving_load_test.py"))
L = importlib.util.module_from_spec(spec)
spec.loader.exec_module(L)


def test_percentile_matches_numpy():
    import numpy as np
    v = [0.3, 0.1, 0.9, 0.5, 0.7, 0.2]
    for q in (0, 50, 90, 95, 100):
        assert abs(L.percentile(v, q) - float(np.percentile(v, q))) < 1e-12


def test_sse_parser_and_summary():
    lines = [b"data: {\"token\": {\"text\": \"a\"}}", b"", b": keep-alive", b"data: not json",
             b"data: {\"token\": {\"text\": \"b\"}, \"generated_text\": \"ab\"}", b"data: [DONE]", b"data: {\"token\": {\"text\": \"x\"}}"]
    evs = list(L.sse_events(lines))
    assert [e["token"]["text"] for e in evs] == ["a", "b"]
    t1, t2 = L.RequestTrace(10.0), L.RequestTrace(10.0)
    for s in (10.5, 10.6, 10.8):
        t1.token(s)
    for s in (11.0, 11.1):
        t2.token(s)
    failed = L.RequestTrace(10.0)
    row = L.summarize([t1, t2, failed], wall=2.0, concurrency=3)
    assert row["requests"] == 3 and row["failed"] == 1 and row["tokens"] == 5
    assert row["first_token_ms_avg"] == 750.0 and row["tokens_per_s"] == 2.5
    assert abs(row["next_token_ms_avg"] - (100 + 200 + 100) / 3) < 1e-2
    assert row["response_s_avg"] == 0.95


def test_run_level_against_fake_stream():
    def send(trace):
        for _ in range(5):
            time.sleep(0.002)
            trace.token(time.perf_counter())
    row = L.run_level(send, concurrency=4, requests_per_client=3)
    assert row["requests"] == 12 and row["failed"] == 0 and row["tokens"] == 60
    assert row["first_token_ms_avg"] >= 2.0 and row["tokens_per_s"] > 0


def test_openai_routes_count_streamed_chunks(monkeypatch):
    """``--api openai`` / ``openai-chat``: one time stamp per streamed chunk with text, on the right route and body"""
    import json
    import requests
    seen = []

    class Resp:
        def __init__(self, lines):
            self.lines = lines

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def raise_for_status(self):
            pass

        def iter_lines(self):
            return iter(self.lines)

    def fake_post(url, json=None, stream=None, timeout=None):
        seen.append((url, json))
        if url.endswith("/v1/chat/completions"):
            chunks = [{"choices": [{"delta": {"role": "assistant"}}]}, {"choices": [{"delta": {"content": "a"}}]},
                      {"choices": [{"delta": {"content": "b"}}]}, {"choices": [{"delta": {}, "finish_reason": "length"}]}]
        else:
            chunks = [{"choices": [{"text": "a"}]}, {"choices": [{"text": "b"}]}, {"choices": [{"text": "c"}]},
                      {"choices": [{"text": "", "finish_reason": "length"}]}]
        return Resp([("data: " + json_dumps(c)).encode() for c in chunks] + [b"data: [DONE]"])

    json_dumps = json.dumps
    monkeypatch.setattr(requests, "post", fake_post)
    t = L.RequestTrace(time.perf_counter())
    L.openai_stream("http://h:1/", "hello", 7, model="m")(t)
    assert len(t.stamps) == 3 and seen[-1][0] == "http://h:1/v1/completions"
    assert seen[-1][1]["max_tokens"] == 7 and seen[-1][1]["stream"] is True and seen[-1][1]["prompt"] == "hello"
    t = L.RequestTrace(time.perf_counter())
    L.openai_stream("http://h:1", "hello", 7, model="m", chat=True)(t)
    assert len(t.stamps) == 2 and seen[-1][0].endswith("/v1/chat/completions")
    assert seen[-1][1]["messages"] == [{"role": "user", "content": "hello"}]
