"""Load test of a running server (FastApp lightweight serving, pipeline / tensor-parallel serving, or any server with
the same ``/generate_stream`` SSE route): for each concurrency level, N clients send prompts back to back for a fixed
number of requests and the script reports average / P90 / P95 first-token latency, next-token latency, response time
and token throughput.

Reference: python/llm/example/GPU/Pipeline-Parallel-Serving/benchmark.py:95-204 (same statistics), and the wrk scripts
wrk_script_1024.lua / docker/llm/serving/xpu/docker/payload-1024.lua (1024-token prompts).

    python benchmark/serving_load_test.py --url http://127.0.0.1:8000 --concurrency 1,4,16,64 --requests-per-client 4 \
        --prompt-tokens 1024 --max-new-tokens 128 --out results.jsonl
"""
from __future__ import annotations

import argparse
import json
import threading
import time
from typing import Callable, Dict, Iterable, List, Optional


def percentile(values: List[float], q: float) -> float:
    """q in [0, 100]; linear interpolation between closest ranks (numpy's default)."""
    if not values:
        return float("nan")
    v = sorted(values)
    pos = (len(v) - 1) * q / 100.0
    lo = int(pos)
    hi = min(lo + 1, len(v) - 1)
    return v[lo] + (v[hi] - v[lo]) * (pos - lo)


class RequestTrace:
    """timestamps of one streamed response"""

    def __init__(self, start: float):
        self.start, self.stamps = start, []

    def token(self, now: float):
        self.stamps.append(now)

    @property
    def first_token(self) -> float:
        return self.stamps[0] - self.start

    @property
    def next_tokens(self) -> List[float]:
        return [b - a for a, b in zip(self.stamps, self.stamps[1:])]

    @property
    def response_time(self) -> float:
        return self.stamps[-1] - self.start


def summarize(traces: List[RequestTrace], wall: float, concurrency: int) -> Dict[str, float]:
    done = [t for t in traces if t.stamps]
    first = [t.first_token for t in done]
    nxt = [d for t in done for d in t.next_tokens]
    resp = [t.response_time for t in done]
    tokens = sum(len(t.stamps) for t in done)
    ms = lambda x: round(x * 1e3, 3)      # noqa: E731
    return {"concurrency": concurrency, "requests": len(traces), "failed": len(traces) - len(done), "tokens": tokens,
            "first_token_ms_avg": ms(sum(first) / max(len(first), 1)), "first_token_ms_p90": ms(percentile(first, 90)),
            "first_token_ms_p95": ms(percentile(first, 95)),
            "next_token_ms_avg": ms(sum(nxt) / max(len(nxt), 1)), "next_token_ms_p90": ms(percentile(nxt, 90)),
            "next_token_ms_p95": ms(percentile(nxt, 95)),
            "response_s_avg": round(sum(resp) / max(len(resp), 1), 4), "response_s_p90": round(percentile(resp, 90), 4),
            "response_s_p95": round(percentile(resp, 95), 4),
            "tokens_per_s": round(tokens / wall, 2) if wall > 0 else 0.0, "wall_s": round(wall, 3)}


def sse_events(lines: Iterable[bytes]):
    """Parsed ``data:`` payloads of a server-sent-event stream; stops at ``[DONE]``."""
    for raw in lines:
        line = raw.decode("utf-8", "replace").strip() if isinstance(raw, bytes) else raw.strip()
        if not line.startswith("data:"):
            continue
        body = line[5:].strip()
        if body == "[DONE]":
            return
        try:
            yield json.loads(body)
        except json.JSONDecodeError:
            continue


def http_stream(url: str, prompt: str, max_new_tokens: int, timeout: float = 600.0) -> Callable[[RequestTrace], None]:
    def run(trace: RequestTrace):
        import requests
        body = {"inputs": prompt, "parameters": {"max_new_tokens": max_new_tokens, "do_sample": False}, "stream": True}
        with requests.post(url.rstrip("/") + "/generate_stream", json=body, stream=True, timeout=timeout) as r:
            r.raise_for_status()
            for ev in sse_events(r.iter_lines()):
                if (ev.get("token") or {}).get("text") is not None:
                    trace.token(time.perf_counter())
    return run


def openai_stream(url: str, prompt: str, max_new_tokens: int, model: str = "default", chat: bool = False,
                  timeout: float = 600.0) -> Callable[[RequestTrace], None]:
    """One streamed ``/v1/completions`` (or ``/v1/chat/completions``) request of an OpenAI-compatible server (the
    built-in one, the vLLM-style entry points, FastChat's): one time stamp per streamed chunk that carries text.
    Reference client: docker/llm/serving/xpu/docker/vllm_online_benchmark.py:90-190."""
    def run(trace: RequestTrace):
        import requests
        if chat:
            body = {"model": model, "messages": [{"role": "user", "content": prompt}], "max_tokens": max_new_tokens,
                    "temperature": 0.0, "stream": True}
            route = "/v1/chat/completions"
        else:
            body = {"model": model, "prompt": prompt, "max_tokens": max_new_tokens, "temperature": 0.0, "stream": True}
            route = "/v1/completions"
        with requests.post(url.rstrip("/") + route, json=body, stream=True, timeout=timeout) as r:
            r.raise_for_status()
            for ev in sse_events(r.iter_lines()):
                for choice in ev.get("choices") or []:
                    text = (choice.get("delta") or {}).get("content") if chat else choice.get("text")
                    if text:
                        trace.token(time.perf_counter())
    return run


def run_level(send: Callable[[RequestTrace], None], concurrency: int, requests_per_client: int) -> Dict[str, float]:
    traces: List[RequestTrace] = []
    lock = threading.Lock()

    def client():
        for _ in range(requests_per_client):
            t = RequestTrace(time.perf_counter())
            try:
                send(t)
            except Exception:      # a failed request is counted, the level goes on
                t.stamps = []
            with lock:
                traces.append(t)

    threads = [threading.Thread(target=client) for _ in range(concurrency)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    return summarize(traces, time.perf_counter() - t0, concurrency)


def synthetic_prompt(n_tokens: int) -> str:
    # one short common word per token for most tokenizers; the server truncates / pads nothing, so the prompt length
    # the model sees is within a few tokens of n_tokens
    return " ".join(["hello"] * n_tokens)


def main(argv: Optional[List[str]] = None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--url", default="http://127.0.0.1:8000")
    ap.add_argument("--concurrency", default="1,2,4,8")
    ap.add_argument("--requests-per-client", type=int, default=4)
    ap.add_argument("--prompt-tokens", type=int, default=1024)
    ap.add_argument("--prompt-file", default=None, help="use this text as the prompt instead of a synthetic one")
    ap.add_argument("--max-new-tokens", type=int, default=128)
    ap.add_argument("--out", default=None)
    ap.add_argument("--api", default="tgi", choices=["tgi", "openai", "openai-chat"],
                    help="tgi: /generate_stream; openai: /v1/completions; openai-chat: /v1/chat/completions (streamed)")
    ap.add_argument("--model", default="default", help="model name sent with the OpenAI routes")
    args = ap.parse_args(argv)
    prompt = open(args.prompt_file).read() if args.prompt_file else synthetic_prompt(args.prompt_tokens)
    if args.api == "tgi":
        send = http_stream(args.url, prompt, args.max_new_tokens)
    else:
        send = openai_stream(args.url, prompt, args.max_new_tokens, model=args.model, chat=args.api == "openai-chat")
    rows = []
    for c in [int(x) for x in args.concurrency.split(",")]:
        row = run_level(send, c, args.requests_per_client)
        rows.append(row)
        print(json.dumps(row), flush=True)
        if args.out:
            with open(args.out, "a") as f:
                f.write(json.dumps(row) + "\n")
    return rows


if __name__ == "__main__":
    main()
