"""``BenchmarkWrapper``: per-token latency and peak-memory measurement around ``generate``.

The reference ships version-specific copies of HuggingFace's whole generation loop with timers inserted
(python/llm/dev/benchmark/benchmark_util.py and python/llm/src/ipex_llm/utils/benchmark_util_4_47.py:3372-3393);
the numbers it publishes are ``first_cost`` (s), ``rest_cost_mean`` (s/token), ``encoder_time``, ``peak_memory`` (GB).
Here the native engine times its own steps; for every other path a forward pre/post hook pair on the model times
each decoding forward (call 0 = prefill = first token) without touching the generation loop.
"""
from __future__ import annotations

import time

import torch


class BenchmarkWrapper:
    def __init__(self, model, do_print: bool = True, verbose: bool = False):
        self.model = model
        self.do_print = do_print
        self.verbose = verbose
        self.encoder_time = 0.0
        self.first_cost = 0.0
        self.rest_cost_mean = 0.0
        self.peak_memory = 0.0
        self.first_token_time = 0.0
        self.last_token_time = []
        if do_print:
            print(self.model.__class__)

    def __getattr__(self, attr):
        if attr in self.__dict__:
            return self.__dict__[attr]
        return getattr(self.model, attr)

    def __call__(self, *args, **kwargs):
        return self.model(*args, **kwargs)

    def _sync(self):
        dev = next(self.model.parameters()).device
        if dev.type == "cuda":
            torch.cuda.synchronize()
        return dev

    def generate(self, *args, **kwargs):
        model = self.model
        times, t_in = [], [0.0]

        def pre(_m, _a, _k=None):
            self._sync()
            t_in[0] = time.perf_counter()

        def post(_m, _a, _o):
            self._sync()
            times.append(time.perf_counter() - t_in[0])

        dev = self._sync()
        if dev.type == "cuda":
            torch.cuda.reset_peak_memory_stats()
        h1 = model.register_forward_pre_hook(pre)
        h2 = model.register_forward_hook(post)
        t0 = time.perf_counter()
        try:
            out = model.generate(*args, **kwargs)
        finally:
            h1.remove()
            h2.remove()
        self._sync()
        total = time.perf_counter() - t0
        if times:                           # HF loop ran: one forward per generated token
            self.first_cost = times[0]
            self.last_token_time = times[1:]
            self.rest_cost_mean = sum(times[1:]) / max(1, len(times) - 1)
        else:                               # native engine / lookup / speculative keep their own clocks
            self.first_cost = float(getattr(model, "first_cost", 0.0) or getattr(model, "first_token_time", 0.0))
            self.rest_cost_mean = float(getattr(model, "rest_cost_mean", 0.0))
            self.last_token_time = list(getattr(model, "last_token_time", []))
        self.first_token_time = self.first_cost
        self.total_time = total
        if dev.type == "cuda":
            self.peak_memory = torch.cuda.max_memory_reserved() / (1024 ** 3)
        if self.do_print:
            print(f"=========First token cost {self.first_cost:.4f} s and {self.peak_memory} GB=========")
            if self.rest_cost_mean:
                print(f"=========Rest tokens cost average {self.rest_cost_mean:.4f} s and {self.peak_memory} GB========")
        return out
