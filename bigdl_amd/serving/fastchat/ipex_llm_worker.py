"""FastChat model worker backed by a low-bit model.

Reference: python/llm/src/ipex_llm/serving/fastchat/ipex_llm_worker.py (``BigDLLLMWorker(BaseModelWorker)``:
``generate_stream_gate`` streams JSON chunks ``{"text", "error_code", "usage", "finish_reason"}`` terminated by a NUL
byte, ``generate_gate`` returns the last chunk; CLI flags ``--low-bit``, ``--device``, ``--speculative``,
``--load-low-bit-model``).  With ``fschat`` installed the class derives from its ``BaseModelWorker`` (controller
registration, heart beat); without it the same generation interface is still usable directly.
"""
from __future__ import annotations

import json
import uuid
from threading import Thread
from typing import List, Optional

import torch

try:
    from fastchat.serve.base_model_worker import BaseModelWorker
    HAVE_FASTCHAT = True
except Exception:
    HAVE_FASTCHAT = False

    class BaseModelWorker:
        def __init__(self, controller_addr, worker_addr, worker_id, model_path, model_names, limit_worker_concurrency,
                     conv_template=None):
            self.controller_addr, self.worker_addr, self.worker_id = controller_addr, worker_addr, worker_id
            self.model_path, self.model_names = model_path, model_names or [model_path.rstrip("/").split("/")[-1]]
            self.limit_worker_concurrency = limit_worker_concurrency
            self.conv_template = conv_template

        def init_heart_beat(self):
            pass

worker_id = str(uuid.uuid4())[:8]


class BigDLLLMWorker(BaseModelWorker):
    def __init__(self, controller_addr: str = "", worker_addr: str = "", worker_id: str = worker_id, model_path: str = "",
                 model_names: Optional[List[str]] = None, limit_worker_concurrency: int = 5, no_register: bool = True,
                 conv_template: str = None, load_in_low_bit: str = "sym_int4", device: str = "cpu",
                 trust_remote_code: bool = False, embed_in_truncate: bool = False, speculative: bool = False,
                 load_low_bit_model: bool = False, stream_interval: int = 4, model=None, tokenizer=None):
        super().__init__(controller_addr, worker_addr, worker_id, model_path, model_names, limit_worker_concurrency,
                         conv_template)
        self.device = "cuda" if device in ("xpu", "gpu") else device
        self.speculative = speculative
        self.stream_interval = stream_interval
        self.embed_in_truncate = embed_in_truncate
        if model is None:
            from ...transformers.loader import load_model
            model, tokenizer = load_model(model_path, self.device, load_in_low_bit, trust_remote_code, speculative,
                                          load_low_bit_model)
        self.model, self.tokenizer = model, tokenizer
        self.context_len = getattr(model.config, "max_position_embeddings", 2048)
        if not no_register:
            self.init_heart_beat()

    def generate_stream_gate(self, params):
        prompt = params["prompt"]
        temperature = float(params.get("temperature", 1.0))
        max_new_tokens = int(params.get("max_new_tokens", 256))
        stop = params.get("stop") or []
        stop = [stop] if isinstance(stop, str) else list(stop)
        echo = bool(params.get("echo", True))
        from transformers import TextIteratorStreamer
        ids = self.tokenizer(prompt, return_tensors="pt").input_ids
        n_prompt = ids.shape[1]
        ids = ids.to(next(p for p in self.model.parameters() if p.numel() > 0).device)
        streamer = TextIteratorStreamer(self.tokenizer, skip_prompt=True, skip_special_tokens=True)
        kw = dict(max_new_tokens=max_new_tokens, streamer=streamer, do_sample=temperature > 1e-5 and temperature != 1.0)
        if kw["do_sample"]:
            kw.update(temperature=temperature, top_p=float(params.get("top_p", 1.0)))
        if params.get("repetition_penalty"):
            kw["repetition_penalty"] = float(params["repetition_penalty"])

        def run():
            with torch.inference_mode():
                self.model.generate(ids, **kw)
        t = Thread(target=run)
        t.start()
        text = prompt if echo else ""
        n = 0
        finish = "length"
        for piece in streamer:
            text += piece
            n += 1
            hit = next((s for s in stop if s and s in text), None)
            if hit is not None:
                text = text[:text.index(hit)]
                finish = "stop"
                break
            if n % self.stream_interval == 0:
                yield json.dumps({"text": text, "error_code": 0, "usage": {"prompt_tokens": n_prompt, "completion_tokens": n,
                                                                           "total_tokens": n_prompt + n},
                                  "finish_reason": None}).encode() + b"\0"
        t.join()
        n_out = len(self.tokenizer(text[len(prompt) if echo else 0:], add_special_tokens=False).input_ids)
        if n_out < max_new_tokens and finish == "length":
            finish = "stop"
        yield json.dumps({"text": text, "error_code": 0, "usage": {"prompt_tokens": n_prompt, "completion_tokens": n_out,
                                                                   "total_tokens": n_prompt + n_out},
                          "finish_reason": finish}).encode() + b"\0"

    def generate_gate(self, params):
        last = None
        for last in self.generate_stream_gate(params):
            pass
        return json.loads(last[:-1].decode())

    @torch.inference_mode()
    def get_embeddings(self, params):
        out = []
        n_tok = 0
        for text in params["input"]:
            ids = self.tokenizer(text, return_tensors="pt").input_ids
            ids = ids.to(next(p for p in self.model.parameters() if p.numel() > 0).device)
            h = self.model.model(input_ids=ids).last_hidden_state[0].float()
            out.append(torch.nn.functional.normalize(h.mean(0), dim=0).cpu().tolist())
            n_tok += ids.shape[1]
        return {"embedding": out, "token_num": n_tok}


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", default="localhost")
    ap.add_argument("--port", type=int, default=21002)
    ap.add_argument("--worker-address", default="http://localhost:21002")
    ap.add_argument("--controller-address", default="http://localhost:21001")
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--model-names", type=lambda s: s.split(","))
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--speculative", action="store_true")
    ap.add_argument("--load-low-bit-model", action="store_true")
    ap.add_argument("--trust-remote-code", action="store_true")
    ap.add_argument("--limit-worker-concurrency", type=int, default=5)
    ap.add_argument("--no-register", action="store_true")
    a = ap.parse_args()
    if not HAVE_FASTCHAT:
        raise SystemExit("fschat is not installed: use `python -m bigdl_amd.serving.fastapi.api_server` for the "
                         "built-in OpenAI / TGI compatible server")
    import uvicorn
    from fastchat.serve import base_model_worker as B
    w = BigDLLLMWorker(a.controller_address, a.worker_address, worker_id, a.model_path, a.model_names,
                       a.limit_worker_concurrency, a.no_register, None, a.low_bit, a.device, a.trust_remote_code,
                       speculative=a.speculative, load_low_bit_model=a.load_low_bit_model)
    B.worker = w
    uvicorn.run(B.app, host=a.host, port=a.port, log_level="info")


if __name__ == "__main__":
    main()
