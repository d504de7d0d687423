"""Offline batched generation with vLLM's call interface, on this package's own serving engine.

The reference's offline entry is ``IPEXLLMClass`` = vLLM's ``LLM`` with ``load_in_low_bit``
(python/llm/src/ipex_llm/vllm/xpu/engine/engine.py:142-246; used as
``llm = LLM(model=..., load_in_low_bit="sym_int4", ...); llm.generate(prompts, SamplingParams(...))`` in
python/llm/example/GPU/vLLM-Serving/offline_inference.py).  When vLLM is installed ``bigdl_amd.vllm.engine`` subclasses
it the same way.  This module is what the same script runs on without vLLM: ``LLM`` loads the model through
``AutoModelForCausalLM`` (low-bit conversion on the host, native decoder engine on the GPU) and ``generate`` pushes all
prompts through ``serving/continuous.py::ContinuousBatcher`` -- iteration-level batching over ``max_num_seqs`` cache
slots, packed prompt admission, chunked prefill -- or, with ``tensor_parallel_size`` > 1 under a launcher, through
``serving/tp.py::TensorParallelBatcher``.  ``SamplingParams`` / ``RequestOutput`` / ``CompletionOutput`` carry the fields
of vLLM's classes that generation here honours; unknown keywords are accepted and ignored so that scripts written for
vLLM run unchanged.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Union

import torch


@dataclass
class SamplingParams:
    n: int = 1
    temperature: float = 1.0
    top_p: float = 1.0
    top_k: int = -1
    max_tokens: Optional[int] = 16
    min_tokens: int = 0
    stop: Optional[Union[str, List[str]]] = None
    stop_token_ids: Optional[List[int]] = None
    ignore_eos: bool = False
    skip_special_tokens: bool = True
    seed: Optional[int] = None
    extra: dict = field(default_factory=dict)

    def __init__(self, n=1, temperature=1.0, top_p=1.0, top_k=-1, max_tokens=16, min_tokens=0, stop=None,
                 stop_token_ids=None, ignore_eos=False, skip_special_tokens=True, seed=None, **extra):
        if n != 1:
            raise ValueError("n > 1 (several samples per prompt) is not supported: submit the prompt n times")
        self.n, self.temperature, self.top_p, self.top_k = n, float(temperature), float(top_p), int(top_k)
        self.max_tokens, self.min_tokens = max_tokens, int(min_tokens)
        self.stop = [stop] if isinstance(stop, str) else list(stop or [])
        self.stop_token_ids = list(stop_token_ids or [])
        self.ignore_eos, self.skip_special_tokens, self.seed = bool(ignore_eos), bool(skip_special_tokens), seed
        self.extra = extra                       # presence / frequency penalties, logprobs, ...: accepted, not applied

    @property
    def greedy(self) -> bool:
        return self.temperature <= 1e-5


@dataclass
class CompletionOutput:
    index: int
    text: str
    token_ids: List[int]
    finish_reason: Optional[str] = None
    cumulative_logprob: Optional[float] = None
    logprobs: Optional[list] = None


@dataclass
class RequestOutput:
    request_id: str
    prompt: Optional[str]
    prompt_token_ids: List[int]
    outputs: List[CompletionOutput]
    finished: bool = True


def _first_stop(text: str, stops: Sequence[str]) -> int:
    cut = -1
    for s in stops:
        i = text.find(s)
        if i >= 0 and (cut < 0 or i < cut):
            cut = i
    return cut


class LLM:
    """``LLM(model, load_in_low_bit="sym_int4", max_num_seqs=.., max_model_len=.., ...)``; ``model`` is a checkpoint path /
    hub id, or an already built (model, tokenizer) pair (tests, notebooks)."""

    _ids = itertools.count()

    def __init__(self, model, tokenizer=None, load_in_low_bit: str = "sym_int4", dtype: str = "float16",
                 tensor_parallel_size: int = 1, max_num_seqs: int = 64, max_model_len: Optional[int] = None,
                 max_num_batched_tokens: Optional[int] = None, trust_remote_code: bool = False, device: Optional[str] = None,
                 seed: int = 0, **ignored):
        from ..transformers.utils import engine_kv_mode
        if isinstance(model, (tuple, list)):
            model, tokenizer = model
        if isinstance(model, str):
            from transformers import AutoTokenizer
            from ..transformers import AutoModelForCausalLM
            path = model
            model = AutoModelForCausalLM.from_pretrained(path, load_in_low_bit=load_in_low_bit,
                                                         trust_remote_code=trust_remote_code, torch_dtype="auto")
            if tokenizer is None:
                tokenizer = AutoTokenizer.from_pretrained(path, trust_remote_code=trust_remote_code)
        self.model, self.tokenizer = model, tokenizer
        dev = "cuda" if (device in (None, "xpu", "cuda", "gpu") and torch.cuda.is_available()) else "cpu"
        if dev == "cuda" and next(model.parameters()).device.type != "cuda":
            self.model = model = model.half().to("cuda")
        self.device = dev
        self.max_model_len = max_model_len or int(getattr(model.config, "max_position_embeddings", 4096))
        self.seed = seed
        self.batcher = None
        from ..models.hf_patch import get_native_engine, native_engine_supported
        if dev == "cuda" and native_engine_supported(model):
            engine = get_native_engine(model)
            kw = dict(max_num_seqs=max_num_seqs, max_seq_len=self.max_model_len, kv_fp8=engine_kv_mode())
            if max_num_batched_tokens:
                kw["max_prefill_tokens"] = max_num_batched_tokens
            if tensor_parallel_size > 1:
                import torch.distributed as dist
                if not dist.is_initialized() or dist.get_world_size() != tensor_parallel_size:
                    raise RuntimeError("tensor_parallel_size > 1: start one process per GPU (torchrun) with that world "
                                       "size; every rank builds the LLM, rank 0's generate() drives the others")
                from ..parallel.tensor_parallel import shard_engine
                from ..serving.tp import TensorParallelBatcher
                engine = shard_engine(engine, dist.get_rank(), tensor_parallel_size)
                self.batcher = TensorParallelBatcher(engine, seed=seed, max_batched_tokens=max_num_batched_tokens or 8192, **kw)
            else:
                from ..serving.continuous import ContinuousBatcher
                self.batcher = ContinuousBatcher(engine, **kw)

    def get_tokenizer(self):
        return self.tokenizer

    # ------------------------------------------------------------------ prompts
    def _token_ids(self, p) -> (Optional[str], List[int]):
        if isinstance(p, dict):
            if "prompt_token_ids" in p:
                return p.get("prompt"), list(p["prompt_token_ids"])
            p = p["prompt"]
        if isinstance(p, str):
            if self.tokenizer is None:
                raise ValueError("text prompts need a tokenizer")
            return p, list(self.tokenizer(p).input_ids)
        return None, [int(t) for t in p]

    def _eos(self, sp: SamplingParams):
        ids = list(sp.stop_token_ids)
        if not sp.ignore_eos:
            e = getattr(getattr(self.model, "generation_config", None), "eos_token_id", None)
            if e is None and self.tokenizer is not None:
                e = self.tokenizer.eos_token_id
            ids += [e] if isinstance(e, int) else list(e or [])
        return ids or None

    def _finish(self, rid, text_prompt, ids, out_ids, reason, sp: SamplingParams) -> RequestOutput:
        text = ""
        if self.tokenizer is not None:
            text = self.tokenizer.decode(out_ids, skip_special_tokens=sp.skip_special_tokens)
            cut = _first_stop(text, sp.stop)
            if cut >= 0:
                text, reason = text[:cut], "stop"
        if reason == "eos":
            reason = "stop"
        return RequestOutput(rid, text_prompt, ids, [CompletionOutput(0, text, list(out_ids), reason)])

    # ------------------------------------------------------------------ generate
    def generate(self, prompts=None, sampling_params: Union[SamplingParams, Sequence[SamplingParams], None] = None,
                 prompt_token_ids: Optional[List[List[int]]] = None, use_tqdm: bool = False, **ignored) -> List[RequestOutput]:
        if prompts is None:
            prompts = prompt_token_ids
        elif prompt_token_ids is not None and isinstance(prompts, (list, tuple)) and prompts and isinstance(prompts[0], str):
            prompts = [{"prompt": p, "prompt_token_ids": t} for p, t in zip(prompts, prompt_token_ids)]
        if isinstance(prompts, (str, dict)) or (prompts and isinstance(prompts[0], int)):
            prompts = [prompts]
        sps = sampling_params if isinstance(sampling_params, (list, tuple)) else [sampling_params or SamplingParams()] * len(prompts)
        if len(sps) != len(prompts):
            raise ValueError("one SamplingParams per prompt, or one for all")
        items = [self._token_ids(p) for p in prompts]
        if self.batcher is not None:
            return self._generate_batched(items, sps)
        return self._generate_module_path(items, sps)

    def _generate_batched(self, items, sps) -> List[RequestOutput]:
        from ..serving.continuous import SeqRequest
        b = self.batcher
        reqs: Dict[str, SeqRequest] = {}
        order = []
        for (text, ids), sp in zip(items, sps):
            rid = f"offline-{next(self._ids)}"
            room = self.max_model_len - len(ids)
            budget = room if sp.max_tokens is None else min(sp.max_tokens, room)
            if budget <= 0:
                raise ValueError(f"prompt of {len(ids)} tokens leaves no room within max_model_len = {self.max_model_len}")
            req = SeqRequest(rid, ids, max_new_tokens=budget, min_new_tokens=sp.min_tokens, eos_token_id=self._eos(sp),
                             do_sample=not sp.greedy, temperature=max(sp.temperature, 1e-5),
                             top_k=max(sp.top_k, 0), top_p=sp.top_p)
            reqs[rid] = req
            order.append((rid, text, ids, sp))
            b.submit(req)
        seeds = {sp.seed for sp in sps if sp.seed is not None}
        if seeds:
            torch.manual_seed(seeds.pop())
            if torch.cuda.is_available():
                torch.cuda.manual_seed(torch.initial_seed())
        # stop strings are checked on the finished text (a request runs to its token budget or end-of-sequence id)
        while b.has_work():
            b.step()
        return [self._finish(rid, text, ids, reqs[rid].output_ids, reqs[rid].finish_reason, sp) for rid, text, ids, sp in order]

    def _generate_module_path(self, items, sps) -> List[RequestOutput]:
        """No engine for this model / device: one ``generate`` call per prompt on the converted modules."""
        outs = []
        for (text, ids), sp in zip(items, sps):
            rid = f"offline-{next(self._ids)}"
            room = self.max_model_len - len(ids)
            budget = room if sp.max_tokens is None else min(sp.max_tokens, room)
            if sp.seed is not None:
                torch.manual_seed(sp.seed)
            kw = dict(max_new_tokens=budget, do_sample=not sp.greedy)
            if sp.min_tokens:
                kw["min_new_tokens"] = sp.min_tokens
            if not sp.greedy:
                kw.update(temperature=sp.temperature, top_p=sp.top_p, top_k=sp.top_k if sp.top_k > 0 else None)
            eos = self._eos(sp)
            if eos:
                kw["eos_token_id"] = eos
            elif sp.ignore_eos:
                kw["eos_token_id"] = None
            x = torch.tensor([ids], device=self.device)
            with torch.no_grad():
                y = self.model.generate(x, **kw)[0, len(ids):].tolist()
            reason = "length" if len(y) >= budget else "stop"
            if eos and y and y[-1] in eos:
                y, reason = y[:-1], "stop"
            outs.append(self._finish(rid, text, ids, y, reason, sp))
        return outs

    def follow(self):
        """Ranks other than 0 of a tensor-parallel group: mirror rank 0's steps until it shuts down."""
        if hasattr(self.batcher, "follow"):
            self.batcher.follow()

    def shutdown(self):
        if hasattr(self.batcher, "shutdown"):
            self.batcher.shutdown()


__all__ = ["LLM", "SamplingParams", "RequestOutput", "CompletionOutput"]
