"""Single-file native models with the llama-cpp-python style call interface.

The reference wraps per-family llama.cpp-style C runtimes through ctypes (``Llama``, ``Gptneox``, ``Bloom``,
``Starcoder``, ``ChatGLM`` in python/llm/src/ipex_llm/ggml/model/*/).  Here one class loads the GGUF file written
by ``llm_convert`` (or by llama.cpp), converts it to the device layout and generates on the native decoder engine
(CPU fallback: the HuggingFace forward over low-bit linears); the family names are kept as aliases.
"""
from __future__ import annotations

import time
import uuid
from typing import Iterator, List, Optional, Union

import torch


class NativeModel:
    def __init__(self, model_path: str, n_ctx: int = 512, n_threads: int = -1, low_bit: Optional[str] = None,
                 device: Optional[str] = None, seed: int = -1, verbose: bool = True, **kwargs):
        from ...transformers import AutoModelForCausalLM
        from ...transformers.gguf.api import qtype_map
        from ...transformers.gguf.gguf import GGUFFileLoader
        self.model_path = model_path
        self.n_ctx_ = n_ctx
        if low_bit is None:
            ft = GGUFFileLoader(model_path).config.get("general.file_type", 2)
            low_bit = qtype_map.get(ft, "sym_int4")
            if low_bit in ("fp16",):
                low_bit = "sym_int4"
        self.model, self.tokenizer = AutoModelForCausalLM.from_gguf(model_path, low_bit=low_bit)
        dev = device or ("cuda" if torch.cuda.is_available() else "cpu")
        if dev != "cpu":
            self.model = self.model.half().to(dev)
        self.device = dev
        if n_threads and n_threads > 0:
            torch.set_num_threads(n_threads)
        if seed is not None and seed >= 0:
            torch.manual_seed(seed)

    # -- llama-cpp-python style helpers ---------------------------------------------------------
    def tokenize(self, text: Union[str, bytes], add_bos: bool = True) -> List[int]:
        if isinstance(text, bytes):
            text = text.decode("utf-8", errors="ignore")
        ids = self.tokenizer(text, add_special_tokens=False)["input_ids"]
        bos = self.tokenizer.bos_token_id
        return ([bos] if add_bos and bos is not None else []) + ids

    def detokenize(self, tokens: List[int]) -> bytes:
        return self.tokenizer.decode(tokens, skip_special_tokens=True).encode("utf-8")

    def n_ctx(self) -> int:
        return self.n_ctx_

    def n_vocab(self) -> int:
        return self.model.config.vocab_size

    def token_eos(self):
        return self.tokenizer.eos_token_id

    def token_bos(self):
        return self.tokenizer.bos_token_id

    def generate(self, tokens: List[int], top_k: int = 40, top_p: float = 0.95, temp: float = 0.8,
                 repeat_penalty: float = 1.1, max_tokens: int = 128, **kwargs) -> Iterator[int]:
        ids = torch.tensor([tokens], device=self.device)
        do_sample = temp is not None and temp > 0
        out = self.model.generate(ids, max_new_tokens=max_tokens, do_sample=do_sample, temperature=temp if do_sample else None,
                                  top_k=top_k if do_sample else None, top_p=top_p if do_sample else None,
                                  repetition_penalty=repeat_penalty)
        for t in out[0, len(tokens):].tolist():
            yield t

    def __call__(self, prompt: str, max_tokens: int = 128, temperature: float = 0.8, top_p: float = 0.95, top_k: int = 40,
                 stop: Optional[List[str]] = None, stream: bool = False, repeat_penalty: float = 1.1, echo: bool = False,
                 **kwargs):
        toks = self.tokenize(prompt)
        t0 = time.time()
        new = list(self.generate(toks, top_k=top_k, top_p=top_p, temp=temperature, repeat_penalty=repeat_penalty,
                                 max_tokens=max_tokens))
        eos = self.token_eos()
        finish = "length"
        if eos is not None and eos in new:
            new = new[:new.index(eos)]
            finish = "stop"
        text = self.detokenize(new).decode("utf-8", errors="ignore")
        for s in stop or []:
            if s in text:
                text, finish = text[:text.index(s)], "stop"
        if echo:
            text = prompt + text
        res = {"id": f"cmpl-{uuid.uuid4()}", "object": "text_completion", "created": int(t0), "model": self.model_path,
               "choices": [{"text": text, "index": 0, "logprobs": None, "finish_reason": finish}],
               "usage": {"prompt_tokens": len(toks), "completion_tokens": len(new), "total_tokens": len(toks) + len(new)}}
        if stream:
            return iter([res])
        return res

    def embed(self, text: str) -> List[float]:
        ids = torch.tensor([self.tokenize(text)], device=self.device)
        with torch.no_grad():
            h = self.model.model(input_ids=ids).last_hidden_state
        return h[0].float().mean(0).cpu().tolist()

    create_completion = __call__


class Llama(NativeModel):
    pass


class Gptneox(NativeModel):
    pass


class Bloom(NativeModel):
    pass


class Starcoder(NativeModel):
    pass


class ChatGLM(NativeModel):
    pass
