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
