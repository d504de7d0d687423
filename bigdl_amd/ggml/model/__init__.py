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
        """Token ids as they are produced (generation runs on a worker thread and hands tokens over a queue, so a
        caller that stops iterating early does not wait for the whole budget)."""
        import queue
        import threading
        ids = torch.tensor([tokens], device=self.device)
        do_sample = temp is not None and temp > 0
        q: "queue.Queue" = queue.Queue()
        done = object()
        cancel = threading.Event()
        # one generation at a time per model: the engine's KV cache, static buffers and decode graph are shared state
        lock = self.__dict__.setdefault("_generate_lock", threading.Lock())

        class _Cancelled(Exception):
            pass

        class _Tokens:                       # the streamer protocol of ``generate``: put(prompt), put(token)..., end()
            first = True

            def put(self, value):
                if cancel.is_set():          # the consumer went away (stop string, abandoned iterator): stop decoding
                    raise _Cancelled()
                if self.first:
                    self.first = False
                    return
                for t in value.reshape(-1).tolist():
                    q.put(t)

            def end(self):
                q.put(done)

        def run():
            with lock:
                try:
                    if cancel.is_set():
                        return
                    self.model.generate(ids, max_new_tokens=max_tokens, do_sample=do_sample,
                                        temperature=temp if do_sample else None, top_k=top_k if do_sample else None,
                                        top_p=top_p if do_sample else None, repetition_penalty=repeat_penalty,
                                        streamer=_Tokens())
                except _Cancelled:
                    pass
                except Exception as e:       # surface the failure to the consumer instead of leaving it blocked
                    q.put(e)
                finally:
                    q.put(done)
        worker = threading.Thread(target=run, daemon=True)
        worker.start()
        try:
            while True:
                item = q.get()
                if item is done:
                    return
                if isinstance(item, Exception):
                    raise item
                yield item
        finally:
            # reached on exhaustion, on an exception and when the consumer closes the generator early (GeneratorExit):
            # the worker sees the flag at its next token and leaves; the next call finds the model idle
            cancel.set()
            worker.join()

    def _completion_chunks(self, prompt, toks, max_tokens, temperature, top_p, top_k, stop, repeat_penalty):
        """(text delta, finish_reason or None, tokens so far) per generated token, honouring EOS and stop strings"""
        eos = self.token_eos()
        new: List[int] = []
        sent = ""
        finish = "length"
        for t in self.generate(toks, top_k=top_k, top_p=top_p, temp=temperature, repeat_penalty=repeat_penalty,
                               max_tokens=max_tokens):
            if eos is not None and t == eos:
                finish = "stop"
                break
            new.append(t)
            text = self.detokenize(new).decode("utf-8", errors="ignore")
            cut = min((text.find(s) for s in (stop or []) if s and s in text), default=-1)
            if cut >= 0:
                if cut > len(sent):
                    yield text[len(sent):cut], None, len(new)
                finish = "stop"
                break
            if text.endswith("\ufffd"):     # half of a multi-byte character: wait for the rest
                continue
            if len(text) > len(sent):
                yield text[len(sent):], None, len(new)
                sent = text
        yield "", finish, len(new)

    def __call__(self, prompt: str, max_tokens: int = 128, temperature: float = 0.8, top_p: float = 0.95, top_k: int = 40,
                 stop: Optional[List[str]] = None, stream: bool = False, repeat_penalty: float = 1.1, echo: bool = False,
                 **kwargs):
        toks = self.tokenize(prompt)
        cid, t0 = f"cmpl-{uuid.uuid4()}", int(time.time())
        chunks = self._completion_chunks(prompt, toks, max_tokens, temperature, top_p, top_k, stop, repeat_penalty)

        def piece(text, finish):
            return {"id": cid, "object": "text_completion", "created": t0, "model": self.model_path,
                    "choices": [{"text": text, "index": 0, "logprobs": None, "finish_reason": finish}]}
        if stream:
            def gen():
                if echo:
                    yield piece(prompt, None)
                for delta, finish, _ in chunks:
                    if delta or finish:
                        yield piece(delta, finish)
            return gen()
        text, finish, n = "", "length", 0
        for delta, fin, n in chunks:
            text += delta
            finish = fin or finish
        res = piece((prompt if echo else "") + text, finish)
        res["usage"] = {"prompt_tokens": len(toks), "completion_tokens": n, "total_tokens": len(toks) + n}
        return res

    def embed(self, text: str) -> List[float]:
        ids = torch.tensor([self.tokenize(text)], device=self.device)
        with torch.no_grad():
            h = self.model.model(input_ids=ids).last_hidden_state
        return h[0].float().mean(0).cpu().tolist()

    def create_embedding(self, input: Union[str, List[str]], model: Optional[str] = None):
        """OpenAI-style embedding response (mean of the last hidden states per input)."""
        inputs = [input] if isinstance(input, str) else list(input)
        data, n = [], 0
        for i, text in enumerate(inputs):
            data.append({"object": "embedding", "embedding": self.embed(text), "index": i})
            n += len(self.tokenize(text))
        return {"object": "list", "data": data, "model": model or self.model_path,
                "usage": {"prompt_tokens": n, "total_tokens": n}}

    def eval(self, tokens: List[int]) -> torch.Tensor:
        """Logits [len(tokens), n_vocab] of one forward pass over ``tokens`` (no cache kept)."""
        with torch.no_grad():
            return self.model(input_ids=torch.tensor([tokens], device=self.device)).logits[0].float().cpu()

    #: how a chat turn is rendered when the tokenizer brings no template of its own (subclasses override)
    chat_format = "{role}: {content}\n"
    chat_suffix = "assistant:"

    def _chat_prompt(self, messages: List[dict]) -> str:
        tpl = getattr(self.tokenizer, "chat_template", None)
        if tpl:
            return self.tokenizer.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
        return "".join(self.chat_format.format(role=m["role"], content=m["content"]) for m in messages) + self.chat_suffix

    def create_chat_completion(self, messages: List[dict], max_tokens: int = 256, temperature: float = 0.2,
                               top_p: float = 0.95, top_k: int = 40, stop: Optional[List[str]] = None, stream: bool = False,
                               repeat_penalty: float = 1.1, **kwargs):
        """OpenAI-style chat completion over the plain completion call."""
        prompt = self._chat_prompt(messages)
        out = self(prompt, max_tokens=max_tokens, temperature=temperature, top_p=top_p, top_k=top_k, stop=stop,
                   stream=stream, repeat_penalty=repeat_penalty)
        if stream:
            def gen():
                first = True
                for c in out:
                    ch = c["choices"][0]
                    delta = {"content": ch["text"]}
                    if first:
                        delta["role"], first = "assistant", False
                    yield {"id": c["id"].replace("cmpl", "chatcmpl"), "object": "chat.completion.chunk", "created": c["created"],
                           "model": c["model"], "choices": [{"index": 0, "delta": delta, "finish_reason": ch["finish_reason"]}]}
            return gen()
        ch = out["choices"][0]
        return {"id": out["id"].replace("cmpl", "chatcmpl"), "object": "chat.completion", "created": out["created"],
                "model": out["model"], "usage": out["usage"],
                "choices": [{"index": 0, "message": {"role": "assistant", "content": ch["text"]},
                             "finish_reason": ch["finish_reason"]}]}

    create_completion = __call__

    # -- the rest of the llama-cpp-python surface the reference exposes (ggml/model/llama/llama.py:305-1420) -----------
    def n_embd(self) -> int:
        return int(self.model.config.hidden_size)

    def token_nl(self) -> int:
        ids = self.tokenizer("\n", add_special_tokens=False)["input_ids"]
        return ids[-1] if ids else -1

    def sample(self, tokens: List[int], top_k: int = 40, top_p: float = 0.95, temp: float = 0.8,
               repeat_penalty: float = 1.1, **kwargs) -> int:
        """The next token after ``tokens``: one forward pass, then the same processors ``generate`` applies."""
        logits = self.eval(tokens)[-1]
        if repeat_penalty and repeat_penalty != 1.0:
            seen = torch.tensor(sorted(set(tokens)), dtype=torch.long)
            v = logits[seen]
            logits[seen] = torch.where(v > 0, v / repeat_penalty, v * repeat_penalty)
        if temp is None or temp <= 0:
            return int(logits.argmax())
        logits = logits / temp
        if top_k and top_k > 0:
            kth = torch.topk(logits, min(top_k, logits.numel())).values[-1]
            logits = logits.masked_fill(logits < kth, float("-inf"))
        probs = torch.softmax(logits, -1)
        if top_p is not None and top_p < 1.0:
            sp, si = torch.sort(probs, descending=True)
            keep = (torch.cumsum(sp, 0) - sp) < top_p
            probs = torch.zeros_like(probs).scatter(0, si[keep], sp[keep])
            probs = probs / probs.sum()
        return int(torch.multinomial(probs, 1))

    def reset(self):
        """Forget the evaluated prefix (the engine keeps no state between calls: every call starts from its prompt)."""
        self._state_tokens = []

    def set_cache(self, cache):
        """Accepted for llama-cpp-python scripts.  The engine's KV cache lives on the GPU and a prompt pass of a few
        thousand tokens costs milliseconds, so completions are not looked up in a host-side prefix cache."""
        self._prefix_cache = cache

    def save_state(self) -> dict:
        return {"tokens": list(getattr(self, "_state_tokens", [])), "n_ctx": self.n_ctx_}

    def load_state(self, state: dict):
        self._state_tokens = list(state.get("tokens", []))

    @staticmethod
    def logits_to_logprobs(logits: List[float]) -> List[float]:
        return torch.log_softmax(torch.as_tensor(logits, dtype=torch.float32), -1).tolist()

    @staticmethod
    def longest_token_prefix(a, b) -> int:
        n = 0
        for x, y in zip(a, b):
            if x != y:
                break
            n += 1
        return n


class LlamaCache:
    """Host-side store keyed by token prefixes with llama-cpp-python's mapping interface (reference
    ggml/model/llama/llama.py:63-107): ``cache[tokens] = state`` / ``cache[tokens]`` returns the state stored under the
    longest prefix of ``tokens``.  Kept so that scripts which build one and pass it to ``set_cache`` run."""

    def __init__(self, capacity_bytes: int = 2 << 30):
        self.capacity_bytes = capacity_bytes
        self._store = {}

    def _longest(self, key):
        best, best_n = None, 0
        for k in self._store:
            n = NativeModel.longest_token_prefix(k, key)
            if n == len(k) and n > best_n:
                best, best_n = k, n
        return best

    def __contains__(self, key) -> bool:
        return self._longest(tuple(key)) is not None

    def __getitem__(self, key):
        k = self._longest(tuple(key))
        if k is None:
            raise KeyError("no stored prefix of these tokens")
        return self._store[k]

    def __setitem__(self, key, value):
        self._store[tuple(key)] = value
        while len(self._store) > 64:
            self._store.pop(next(iter(self._store)))


def _family(name: str, architectures, chat_format=None, chat_suffix=None):
    """A family class in the reference's naming: refuses files of another architecture instead of mis-loading them."""

    def __init__(self, model_path: str, *args, **kwargs):
        from ...transformers.gguf.gguf import GGUFFileLoader
        arch = GGUFFileLoader(model_path).config.get("general.architecture")
        if arch not in architectures:
            raise ValueError(f"{name} loads {' / '.join(architectures)} files; {model_path} is a {arch} file")
        NativeModel.__init__(self, model_path, *args, **kwargs)
    ns = {"__init__": __init__, "__doc__": f"{name}: single-file native model of the {' / '.join(architectures)} family."}
    if chat_format:
        ns["chat_format"], ns["chat_suffix"] = chat_format, chat_suffix
    return type(name, (NativeModel,), ns)


Llama = _family("Llama", ("llama", "mistral", "qwen2", "baichuan"), "[INST] {content} [/INST]", "")
Gptneox = _family("Gptneox", ("gptneox", "gpt_neox"))
Bloom = _family("Bloom", ("bloom",))
Starcoder = _family("Starcoder", ("starcoder", "gpt_bigcode", "starcoder2"))
ChatGLM = _family("ChatGLM", ("chatglm", "glm"), "<|{role}|>\n{content}", "<|assistant|>\n")
