"""Client of the TGI-style front end (``python -m bigdl_amd.serving.fastchat.tgi_api_server`` over FastChat workers, or the
``/generate`` / ``/generate_stream`` routes of the lightweight FastAPI server): one blocking call and one streamed call.

    python tgi_client.py --url http://localhost:8000 --prompt "What is AI?"
"""
import argparse
import json

import requests

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--url", default="http://localhost:8000")
    ap.add_argument("--prompt", default="What is AI?")
    ap.add_argument("--max-new-tokens", type=int, default=64)
    ap.add_argument("--api-key", default=None)
    a = ap.parse_args()
    headers = {"Authorization": f"Bearer {a.api_key}"} if a.api_key else {}
    body = {"inputs": a.prompt, "parameters": {"max_new_tokens": a.max_new_tokens, "do_sample": False}}
    r = requests.post(a.url + "/generate", json=body, headers=headers, timeout=300)
    r.raise_for_status()
    print("generate:", r.json()["generated_text"])
    print("stream  : ", end="", flush=True)
    with requests.post(a.url + "/generate_stream", json=body, headers=headers, stream=True, timeout=300) as r:
        for line in r.iter_lines():
            if not line.startswith(b"data: ") or line == b"data: [DONE]":
                continue
            ev = json.loads(line[6:])
            print(ev.get("token", {}).get("text", ""), end="", flush=True)
    print()
