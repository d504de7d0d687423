"""OpenAI / TGI compatible server on one GPU
(reference: python/llm/example/GPU/Lightweight-Serving/lightweight_serving.py)."""
import argparse

from ipex_llm.serving.fastapi.api_server import serve

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--port", type=int, default=8000)
    a = ap.parse_args()
    serve(a.repo_id_or_model_path, a.low_bit, a.port)
