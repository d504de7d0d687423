"""Self-speculative decoding: a sym_int4 copy of the model drafts, the fp16/bf16 (or higher-bit) target verifies
a whole draft in one multi-token step.

Capability map to the reference python/llm/src/ipex_llm/transformers/speculative.py: generate front door :51-118,
``clear_benchmarks`` :444-456, ``speculative_generate`` :802-1269 (draft loop with ``th_stop_draft`` early stop,
greedy and sampled verification :1162-1195, KV rollback :1201-1206, adaptive threshold :1225-1236, ``hf_adjust``
:1238-1242, the per-step statistics ``draft_time`` / ``verify_time`` / ``draft_num`` / ``accept_num`` /
``n_drafted`` / ``n_matched``).

Both models run through the stepper interface (transformers/stepper.py).  On the GPU each has its own native
engine and fixed-capacity KV cache; the target verifies ``1 + n_draft`` tokens with one graph replay and a rejected
tail costs one counter write per cache, where the reference re-slices every layer's cache tensors.
"""
from __future__ import annotations

import time
import types
from typing import Optional

import torch

from .stepper import make_stepper


def clear_benchmarks(self):
    self.first_token_time = 0
    self.generate_time = []
    self.draft_time = []
    self.verify_time = []
    self.draft_num = []
    self.accept_num = []
    self.n_drafted = 0
    self.n_matched = 0


def _probs(logits, temperature, top_k, top_p):
    """Filtered sampling distribution (same filters for draft and target so acceptance ratios are consistent)."""
    if temperature and temperature != 1.0:
        logits = logits / temperature
    if top_k and top_k > 0:
        kth = torch.topk(logits, min(top_k, logits.shape[-1]), dim=-1).values[..., -1:]
        logits = logits.masked_fill(logits < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        srt, idx = torch.sort(logits, descending=True, dim=-1)
        p = torch.softmax(srt, -1)
        drop = p.cumsum(-1) - p > top_p
        srt = srt.masked_fill(drop, float("-inf"))
        logits = torch.full_like(logits, float("-inf")).scatter(-1, idx, srt)
    return torch.softmax(logits, -1)


def _engine_pair(model, draft_model, do_sample, kwargs):
    """The engine-level loop for (model, draft_model), or None when this call has to take the generic stepper loop
    (caller-provided steppers, a model without a native engine, CPU)."""
    import os
    skip = kwargs.pop("draft_layers", None) or os.environ.get("BIGDL_SPEC_DRAFT_LAYERS")      # (engine pairs only)
    if "stepper" in kwargs or "draft_stepper" in kwargs or os.environ.get("BIGDL_SPEC_STEPPER_LOOP") == "1":
        return None
    try:
        dev = next(model.parameters()).device
    except (StopIteration, AttributeError):
        return None
    if dev.type != "cuda":
        return None
    from ..models.hf_patch import get_native_engine, native_engine_supported
    if not (native_engine_supported(model) and native_engine_supported(draft_model)):
        return None
    pair = getattr(model, "_spec_engine", None)
    target_engine, draft_engine = get_native_engine(model), get_native_engine(draft_model)
    # draft_layers (keyword of generate, or BIGDL_SPEC_DRAFT_LAYERS): the draft runs only that many of its layers
    # (an int: first / last two + evenly spaced ones; a list: exactly those) on the target's cache rows of those layers
    if isinstance(skip, str):
        skip = [int(x) for x in skip.split(",")] if "," in skip else int(skip)
    key = tuple(skip) if isinstance(skip, (list, tuple)) else skip
    if pair is None or pair.target is not target_engine or getattr(pair, "full_draft", pair.draft) is not draft_engine \
            or getattr(pair, "skip_key", None) != key:
        # a different draft model (or a rebuilt engine) gets its own pair; the old pair's buffers go with it
        from ..models.spec_engine import SelfSpeculativeEngine, default_kept_layers, layer_skip_draft_of
        runs = draft_engine
        if skip:
            kept = list(skip) if isinstance(skip, (list, tuple)) else default_kept_layers(len(draft_engine.layers), skip)
            if len(kept) < len(draft_engine.layers):
                runs = layer_skip_draft_of(draft_engine, kept)
        pair = SelfSpeculativeEngine(target_engine, runs)
        pair.full_draft, pair.skip_key = draft_engine, key
        object.__setattr__(model, "_spec_engine", pair)
    return pair


@torch.no_grad()
def speculative_generate(self, inputs: Optional[torch.Tensor] = None, draft_model=None, max_new_tokens=10,
                         max_step_draft=8, th_stop_draft=0.8, auto_th_stop_draft=True,
                         auto_parameters=(1, 0.5, 0.9, 1e-2, 0.9), hf_adjust=False, min_step_draft=3,
                         generation_config=None, attention_mask=None, streamer=None, **sampling_kwargs):
    input_ids = sampling_kwargs.pop("input_ids", inputs)
    assert input_ids is not None and input_ids.shape[0] == 1, "speculative decoding supports batch size 1"
    draft_model = draft_model if draft_model is not None else getattr(self, "draft_model", None)
    assert draft_model is not None, "speculative decoding needs a draft model (from_pretrained(..., speculative=True))"
    gc = generation_config or getattr(self, "generation_config", None)
    do_sample = bool(sampling_kwargs.get("do_sample", getattr(gc, "do_sample", False)))
    temperature = sampling_kwargs.get("temperature", 1.0) or 1.0
    top_k = sampling_kwargs.get("top_k", 0) or 0
    top_p = sampling_kwargs.get("top_p", 1.0) or 1.0
    eos = sampling_kwargs.get("eos_token_id", getattr(gc, "eos_token_id", None))
    eos = set([eos] if isinstance(eos, int) else (eos or []))
    auto_parameters = list(auto_parameters)

    fast = _engine_pair(self, draft_model, do_sample, sampling_kwargs)
    if fast is not None:
        # both models run on the native engine: shared KV cache, device-chained draft loop, one host read per round,
        # draft confidence read one step late from pinned memory (bigdl_amd/models/spec_engine.py)
        from .utils import engine_kv_mode
        fast.max_step_draft = max_step_draft
        fast.min_step_draft = max(1, min_step_draft)
        out = fast.generate(input_ids, max_new_tokens=max_new_tokens, eos_token_id=eos, kv_fp8=engine_kv_mode(),
                            streamer=streamer, do_sample=do_sample, temperature=temperature, top_k=top_k, top_p=top_p,
                            th_stop_draft=th_stop_draft, auto_th_stop_draft=auto_th_stop_draft,
                            auto_parameters=auto_parameters)
        for k in ("first_token_time", "generate_time", "draft_time", "verify_time", "draft_num", "accept_num", "n_drafted",
                  "n_matched", "n_token_generated", "e2e_time_without_first", "first_cost", "rest_cost_mean"):
            object.__setattr__(self, k, getattr(fast, k))
        return out

    cap = input_ids.shape[1] + max_new_tokens + max_step_draft + 16
    target = sampling_kwargs.pop("stepper", None) or make_stepper(self, capacity=cap)
    draft = sampling_kwargs.pop("draft_stepper", None) or make_stepper(draft_model, capacity=cap)
    dev = target.device
    clear_benchmarks(self)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    def pick(logits):
        if do_sample:
            return torch.multinomial(_probs(logits, temperature, top_k, top_p), 1).squeeze(-1)
        return logits.argmax(-1)

    # ---- first token: target prefill (the draft cache is filled from the same prompt)
    sync()
    tic = time.perf_counter()
    logits = target.prefill(input_ids)
    cur = int(pick(logits)[0])
    sync()
    self.first_token_time = time.perf_counter() - tic
    draft.prefill(input_ids)
    generated = [cur]
    if streamer is not None:
        streamer.put(input_ids.cpu())
        streamer.put(torch.tensor([cur]))
    e2e_tic = time.perf_counter()
    tmp_matchness = 0.0
    step_verify = 0
    done = cur in eos
    draft_gen_length = max_step_draft + 6 if hf_adjust else max_step_draft + 1
    # invariant at loop top: both caches hold prompt + generated[:-1]; generated[-1] is not in any cache
    while not done and len(generated) < max_new_tokens:
        base = target.length
        # ---- draft
        t0 = time.perf_counter()
        drafts, draft_probs = [], []
        tok = generated[-1]
        room = max_new_tokens - len(generated) - 1
        for i in range(min(max_step_draft, max(room, 0))):
            dl = draft.forward(torch.tensor([[tok]], dtype=torch.long, device=draft.device))[0, -1]
            if do_sample:
                p = _probs(dl, temperature, top_k, top_p)
                tok = int(torch.multinomial(p, 1))
                conf = float(p[tok])
                draft_probs.append(p)
            else:
                p = torch.softmax(dl, -1)
                conf, t = p.max(-1)
                tok, conf = int(t), float(conf)
            drafts.append(tok)
            if tok in eos:
                break
            # stop drafting when the draft model is unsure (after min_step_draft tokens)
            if conf < th_stop_draft and (i + 1) >= min_step_draft:
                break
        sync()
        t1 = time.perf_counter()
        self.draft_time.append(t1 - t0)
        n_d = len(drafts)
        self.draft_num.append(n_d)
        # ---- verify: one target step over [last accepted, drafts...]
        tokens = torch.tensor([[generated[-1]] + drafts], dtype=torch.long, device=dev)
        out = target.forward(tokens)[0]                                  # [1+n_d, V]
        if do_sample and n_d:
            tp = _probs(out, temperature, top_k, top_p)                  # target probabilities per position
            dp = torch.stack(draft_probs).to(tp.device)
            idx = torch.arange(n_d, device=tp.device)
            dt = torch.tensor(drafts, device=tp.device)
            ratio = (tp[idx, dt] / dp[idx, dt].clamp_min(1e-20)).clamp(max=1.0)
            rejected = (torch.rand(n_d, device=tp.device) > ratio).nonzero()
            if rejected.numel() == 0:
                m = n_d
                nxt = int(torch.multinomial(tp[n_d], 1))
            else:
                m = int(rejected[0])
                resid = (tp[m] - dp[m]).clamp_min(0)
                s = resid.sum()
                nxt = int(torch.multinomial(resid / s if float(s) > 0 else tp[m], 1))
            accepted = drafts[:m] + [nxt]
        else:
            choice = pick(out).tolist()
            m = 0
            while m < n_d and drafts[m] == choice[m]:
                m += 1
            accepted = choice[: m + 1]
        sync()
        t2 = time.perf_counter()
        self.verify_time.append(t2 - t1)
        self.generate_time.append(t2 - t0)
        self.accept_num.append(m + 1)
        self.n_matched += m
        self.n_drafted += n_d
        # ---- roll both caches back to prompt + generated + accepted[:-1]
        keep = base + 1 + m
        target.crop(keep)
        if draft.length > keep:
            draft.crop(keep)
        elif draft.length < keep:
            # every draft was accepted: the draft cache misses the last draft token
            miss = ([generated[-1]] + drafts)[draft.length - base: keep - base]
            draft.forward(torch.tensor([miss], dtype=torch.long, device=draft.device))
        if eos:
            for i, t in enumerate(accepted):
                if t in eos:
                    accepted, done = accepted[: i + 1], True
                    break
        generated.extend(accepted)
        if streamer is not None:
            streamer.put(torch.tensor(accepted))
        step_verify += 1
        if auto_th_stop_draft and n_d > 0 and step_verify % auto_parameters[0] == 0:
            tmp_matchness = auto_parameters[1] * tmp_matchness + (1 - auto_parameters[1]) * (m / n_d)
            if tmp_matchness < auto_parameters[2]:
                new_th = th_stop_draft + auto_parameters[3]
            elif n_d == max_step_draft:
                new_th = th_stop_draft
            else:
                new_th = th_stop_draft - auto_parameters[3]
            th_stop_draft = auto_parameters[4] * th_stop_draft + (1 - auto_parameters[4]) * new_th
        if hf_adjust:
            if m == max_step_draft:
                max_step_draft = min(draft_gen_length - 1, max_step_draft + 1)
            else:
                max_step_draft = max(1, max_step_draft - 1)
    if streamer is not None:
        streamer.end()
    sync()
    generated = generated[:max_new_tokens]
    self.n_token_generated = len(generated)
    self.e2e_time_without_first = time.perf_counter() - e2e_tic
    self.first_cost = self.first_token_time
    self.rest_cost_mean = self.e2e_time_without_first / max(1, len(generated) - 1)
    self.th_stop_draft_final = th_stop_draft
    return torch.cat([input_ids.to(dev), torch.tensor([generated], dtype=input_ids.dtype, device=dev)], 1)


def attach_speculative(model, draft_model):
    """Give `model` a ``draft_model`` plus ``speculative_generate`` / ``clear_benchmarks``; ``generate`` then
    routes batch-1 calls through the speculative loop (see bigdl_amd/models/hf_patch.py ``native_generate``)."""
    model.draft_model = draft_model      # a submodule, so ``model.to(device)`` moves both
    model.speculative_generate = types.MethodType(speculative_generate, model)
    model.clear_benchmarks = types.MethodType(clear_benchmarks, model)
    from ..models.hf_patch import install_generate_front_door
    install_generate_front_door(model)
    return model
