"""Self-speculative decoding on the native engine: a low-bit copy of the model drafts, the fp16 (or higher-bit) target
verifies the whole draft in one multi-row step.

Capability: the reference's ``speculative_generate`` (python/llm/src/ipex_llm/transformers/speculative.py:802-1269):
draft loop :944-1052, the draft runs on the target's own ``past_key_values`` :931-935, greedy verification :1162-1175,
KV rollback :1201-1206, ``max_step_draft`` adaptation (``hf_adjust``) :1238-1242, statistics :444-456.

How it maps to this engine:

* ONE KV cache.  The draft engine is handed the target's cache object: it attends to the target's exact prompt keys
  (no draft prefill, no second cache) and appends its own approximate keys for the tokens it is drafting; the
  verification step then rewrites exactly those positions with the target's keys.  Rejection is a counter write.
* The draft loop is a chain of graph replays: the greedy token of step i is written by the graph where step i + 1
  reads its input, a tiny device copy files it in the round's draft buffer.  Nothing returns to the host per draft
  token.
* Verification is one graph replay over ``k + 1`` rows; the accepted prefix length is computed on the device and
  ``[m, target tokens]`` come back in ONE host read per round.
* The draft length adapts between rounds from what was accepted (host side, no extra read): a fully accepted round
  grows it, a rejection shrinks it.  With ``th_stop_draft`` the draft's confidence decides instead: it is computed on
  the device behind every draft step and read by the host one step late, from pinned memory, so drafting stops
  without the device ever waiting for the host (reference :1046-1052; ``auto_th_stop_draft`` :1225-1236).
* Sampling: draft distributions stay on the device, acceptance (min(1, q / p)), the first rejection and the
  replacement sample from max(q - p, 0) are computed there too (reference :1162-1188).
"""
from __future__ import annotations

import time
from typing import Optional

import torch


class SelfSpeculativeEngine:
    def __init__(self, target, draft, max_step_draft: int = 6, min_step_draft: int = 1, adapt: bool = True):
        assert target.cfg.num_layers == draft.cfg.num_layers and target.cfg.kv_size == draft.cfg.kv_size \
            and target.cfg.head_dim == draft.cfg.head_dim, "draft and target must share the KV layout"
        self.target, self.draft = target, draft
        self.max_step_draft, self.min_step_draft, self.adapt = max_step_draft, max(1, min_step_draft), adapt
        self.device = target.device
        self.clear_benchmarks()

    def clear_benchmarks(self):
        self.first_token_time = 0.0
        self.generate_time, self.draft_time, self.verify_time = [], [], []
        self.draft_num, self.accept_num = [], []
        self.n_drafted = self.n_matched = 0
        self.first_cost = self.rest_cost_mean = 0.0
        self.n_token_generated = 0

    def _share_cache(self, batch, capacity, kv_fp8):
        t, d = self.target, self.draft
        t.ensure_cache(batch, capacity, fp8=kv_fp8)
        if d.cache is not t.cache:
            d.cache = t.cache
            d._graphs.clear()
            d._static.clear()
            d.inv_freq, d.rope_scaling_factor = t.inv_freq, t.rope_scaling_factor
            d.rope_table, d.no_rope, d.fused_rope = t.rope_table, t.no_rope, t.fused_rope
            d._rope_shift_buf = None
            d._set_rope_shift(0)
        t.cache.reset()

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, max_new_tokens: int = 32, eos_token_id=None, kv_fp8=False, streamer=None,
                 time_phases: bool = False, do_sample: bool = False, temperature: float = 1.0, top_k: int = 0,
                 top_p: float = 1.0, th_stop_draft: Optional[float] = None, auto_th_stop_draft: bool = False,
                 auto_parameters=(1, 0.5, 0.9, 1e-2, 0.9), stop_lag: int = 1, oracle_accept: Optional[float] = None,
                 **unused):
        """Greedy or sampled speculative generation.

        ``th_stop_draft``: stop drafting once the draft's confidence in a token it produced (the probability of the
        token it picked) falls below the threshold and ``min_step_draft`` tokens exist.  The confidence of every draft
        step is computed on the device right behind the step's graph and lands in pinned host memory by an
        asynchronous copy; the host looks at step ``i - stop_lag``'s value before it queues step ``i + 1``.  With the
        default lag of one step that value is already there (its step finished while the next one was running), so the
        device never waits for the host and the loop stops at most one token late; ``stop_lag=0`` stops exactly where
        the reference does and pays one host round trip per draft token.  ``auto_th_stop_draft`` moves the threshold
        from the acceptance history with the reference's rule (``auto_parameters``).

        ``do_sample``: the draft samples from its filtered distribution, the target accepts token i with probability
        min(1, q_i / p_i) and replaces the first rejected token by a sample of normalize(max(q - p, 0)) (or, when all
        were accepted, adds a sample of its own next distribution); everything up to that token is decided on the
        device and one host read per round returns it.

        ``oracle_accept`` (measurement only): ignore the comparison and accept each draft token with this probability
        (1.0: every round accepts all k), so that speed can be tabulated against acceptance rate; the text produced
        is meaningless.

        ``time_phases`` synchronises between the draft and verify phases of every round to fill ``draft_time`` /
        ``verify_time`` (costs one extra host wait per round); otherwise only ``generate_time`` per round is
        recorded."""
        t, d, dev = self.target, self.draft, self.device
        input_ids = input_ids.to(dev)
        assert input_ids.shape[0] == 1, "speculative decoding serves one sequence"
        s = input_ids.shape[1]
        kmax = self.max_step_draft
        self._share_cache(1, s + max_new_tokens + kmax + 2, kv_fp8)
        cache = t.cache
        eos = set([eos_token_id] if isinstance(eos_token_id, int) else (eos_token_id or []))
        self.clear_benchmarks()
        auto = list(auto_parameters)
        matchness = 0.0
        self.th_stop_draft = th = th_stop_draft
        vocab = t.cfg.vocab_size

        def dist_of(logits):
            return _filtered_probs(logits.float(), temperature, top_k, top_p)

        torch.cuda.synchronize()
        tic = time.perf_counter()
        logits = t.prefill(input_ids)[:, -1, :]
        if d._rope_shift != t._rope_shift:                   # a compressed prompt: both count positions from its real length
            d._set_rope_shift(t._rope_shift)
        last = (torch.multinomial(dist_of(logits), 1) if do_sample else logits.argmax(-1)).view(1, 1)      # device
        generated = [int(last)]
        self.first_token_time = self.first_cost = time.perf_counter() - tic
        if streamer is not None:
            streamer.put(input_ids.cpu())
            streamer.put(last.view(-1).cpu())
        drafts = torch.zeros(kmax, dtype=torch.int64, device=dev)
        verify_in = torch.zeros(1, kmax + 1, dtype=torch.int64, device=dev)
        result = torch.zeros(kmax + 2, dtype=torch.int64, device=dev)      # [m, target tokens ...]
        watch = th is not None
        if watch:
            conf_dev = torch.zeros(kmax, dtype=torch.float32, device=dev)
            conf_host = torch.zeros(kmax, dtype=torch.float32).pin_memory()
            events = [torch.cuda.Event() for _ in range(kmax)]
        if do_sample:
            draft_probs = torch.zeros(kmax + 1, vocab, dtype=torch.float32, device=dev)   # row k stays zero (see below)
        k = min(kmax, max(self.min_step_draft, 4)) if not watch else kmax
        done = generated[-1] in eos
        e2e = time.perf_counter()
        rounds = 0
        while not done and len(generated) < max_new_tokens:
            t0 = time.perf_counter()
            base = cache.host_len
            k = max(1, min(k, max_new_tokens - len(generated) - 1)) if max_new_tokens - len(generated) > 1 else 0
            # ---- draft: up to k chained replays of the low-bit engine on the shared cache
            n = 0
            while n < k:
                if watch and n - 1 - stop_lag >= 0:
                    j = n - 1 - stop_lag                      # newest step whose confidence the host may look at
                    events[j].synchronize()
                    if j + 1 >= self.min_step_draft and float(conf_host[j]) < th:
                        break
                if do_sample:
                    lg, _ = d.decode(last if n == 0 else drafts[n - 1:n].view(1, 1), 1, chain=False)
                    pr = dist_of(lg.view(1, -1))
                    draft_probs[n].copy_(pr[0])
                    drafts[n:n + 1].copy_(torch.multinomial(pr, 1).view(-1))
                    if watch:
                        conf_dev[n:n + 1].copy_(pr[0].gather(0, drafts[n:n + 1]))
                else:
                    lg, chained = d.decode(last if n == 0 else None, 1, chain=True)
                    drafts[n:n + 1].copy_(chained)
                    if watch:       # probability of the greedy token: 1 / sum(exp(logit - max))
                        row = lg.view(-1)
                        conf_dev[n:n + 1].copy_(torch.exp(row - row.max()).sum().reciprocal().view(1))
                if watch:
                    conf_host[n:n + 1].copy_(conf_dev[n:n + 1], non_blocking=True)
                    events[n].record()
                n += 1
            k = n
            if time_phases:
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            # ---- verify: [last, drafts] in one step of the target, over the positions the draft just used
            cache.crop(base)
            verify_in[0, 0:1].copy_(last.view(-1))
            if k:
                verify_in[0, 1:k + 1].copy_(drafts[:k])
            vlogits, choice = t.decode(verify_in[:, :k + 1], k + 1, chain=False)     # choice: greedy tokens [k + 1]
            if do_sample:
                q_all = dist_of(vlogits.view(k + 1, -1))                             # target distributions, rows 0..k
                if k:
                    tok = drafts[:k]
                    p = draft_probs[:k].gather(1, tok[:, None]).squeeze(1)
                    q = q_all[:k].gather(1, tok[:, None]).squeeze(1)
                    ok = (torch.rand(k, device=dev) <= q / p.clamp_min(1e-30)).to(torch.int64)
                    m_dev = torch.cumprod(ok, 0).sum()
                else:
                    m_dev = torch.zeros((), dtype=torch.int64, device=dev)
                # the token after the accepted prefix: from max(q_m - p_m, 0) at the first rejection; row k of the
                # draft table is all zero, so "everything accepted" (m = k) samples the target's own next distribution
                draft_probs[k].zero_()
                resid = (q_all.index_select(0, m_dev.view(1)) - draft_probs[:k + 1].index_select(0, m_dev.view(1))).clamp_min(0)
                resid = torch.where(resid.sum() > 0, resid, q_all.index_select(0, m_dev.view(1)))
                bonus = torch.multinomial(resid / resid.sum(), 1).view(-1)
                choice = torch.cat([drafts[:k], bonus])                               # [k + 1]; only [:m] + bonus are used
                choice = torch.where(torch.arange(k + 1, device=dev) == m_dev, bonus.expand(k + 1), choice)
            elif k:
                ok = (choice[:k] == drafts[:k]).to(torch.int64)
                m_dev = torch.cumprod(ok, 0).sum()
            else:
                m_dev = torch.zeros((), dtype=torch.int64, device=dev)
            if oracle_accept is not None and k:
                ok = (torch.rand(k, device=dev) < oracle_accept).to(torch.int64)
                m_dev = torch.cumprod(ok, 0).sum()
            result[0] = m_dev
            result[1:k + 2].copy_(choice)
            host = result[:k + 2].tolist()                                           # the round's one host read
            m, choice_h = host[0], host[1:]
            accepted = choice_h[:m + 1]
            cache.crop(base + 1 + m)
            last = choice[m:m + 1].view(1, 1).clone()
            t2 = time.perf_counter()
            self.draft_time.append(t1 - t0)
            self.verify_time.append(t2 - t1)
            self.generate_time.append(t2 - t0)
            self.draft_num.append(k)
            self.accept_num.append(m + 1)
            self.n_drafted += k
            self.n_matched += m
            rounds += 1
            if eos:
                for i, tok_h in enumerate(accepted):
                    if tok_h in eos:
                        accepted, done = accepted[:i + 1], True
                        break
            generated.extend(accepted)
            if streamer is not None:
                streamer.put(torch.tensor(accepted))
            if watch:
                if auto_th_stop_draft and k and rounds % max(1, int(auto[0])) == 0:
                    # reference speculative.py:1225-1236: unsure drafts raise the threshold, confident ones lower it
                    matchness = auto[1] * matchness + (1 - auto[1]) * (m / k)
                    if matchness < auto[2]:
                        new_th = th + auto[3]
                    else:
                        new_th = th if k == kmax else th - auto[3]
                    th = auto[4] * th + (1 - auto[4]) * new_th
                    self.th_stop_draft = th
                k = kmax                                     # the threshold decides the length of the next draft
            elif self.adapt and k:
                k = min(kmax, k + 1) if m == k else max(self.min_step_draft, k - 1)
        if streamer is not None:
            streamer.end()
        generated = generated[:max_new_tokens]
        self.n_token_generated = len(generated)
        self.e2e_time_without_first = time.perf_counter() - e2e
        self.rest_cost_mean = self.e2e_time_without_first / max(1, len(generated) - 1)
        return torch.cat([input_ids, torch.tensor([generated], dtype=input_ids.dtype, device=dev)], 1)


def _filtered_probs(logits, temperature, top_k, top_p):
    """Sampling distribution [rows, V] after temperature / top-k / top-p (the same filters for draft and target, so the
    acceptance ratio q / p compares like with like)."""
    if temperature and temperature != 1.0:
        logits = logits / temperature
    if top_k and top_k > 0:
        kth = torch.topk(logits, min(top_k, logits.shape[-1]), dim=-1).values[..., -1:]
        logits = logits.masked_fill(logits < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        srt, idx = torch.sort(logits, descending=True, dim=-1)
        pr = torch.softmax(srt, -1)
        srt = srt.masked_fill(pr.cumsum(-1) - pr > top_p, float("-inf"))
        logits = torch.full_like(logits, float("-inf")).scatter(-1, idx, srt)
    return torch.softmax(logits, -1)


def quantized_draft_of(target, qtype, lm_head_qtype: Optional[int] = None):
    """A draft engine for ``target``: the same weights quantized to ``qtype`` (embedding and norms are shared, not
    copied).  This is what ``from_pretrained(..., speculative=True)`` builds from a checkpoint (reference
    transformers/model.py:330-340: the draft is the sym_int4 conversion of the same model)."""
    import dataclasses
    from ..ops import host
    from .decoder import Lin, NativeDecoder
    from ..transformers.low_bit_linear import FP4Params

    def q(lin, qt=qtype):
        if lin is None:
            return None
        # host quantizer (csrc/host/quantize.cpp), then the tiled device layout
        raw = host.quantize(lin.dense().float().cpu(), qt)
        data = host.relayout_to_device(raw, qt, lin.n, lin.k).to(target.device)
        return Lin(FP4Params(data, quantized=True, _shape=(lin.n, lin.k), qtype=qt, qlayout="t16"), bias=lin.bias)

    layers = []
    for lw in target.layers:
        assert lw.moe is None and lw.mla is None, "quantized_draft_of covers dense decoder layers"
        layers.append(dataclasses.replace(lw, qkv=q(lw.qkv), o=q(lw.o), gate=q(lw.gate), up=q(lw.up), down=q(lw.down)))
    draft = NativeDecoder(target.cfg, target.embed, layers, target.final_norm, q(target.lm_head, lm_head_qtype or qtype),
                          device=target.device, final_norm_b=target.final_norm_b, pos_embed=target.pos_embed,
                          embed_norm=target.embed_norm)
    return draft


def default_kept_layers(num_layers: int, keep: int):
    """Which ``keep`` of ``num_layers`` layers a layer-skipping draft runs: the first two and the last two (the layers
    that turn embeddings into features and features into logits) plus evenly spaced ones in between."""
    keep = max(1, min(keep, num_layers))
    if keep >= num_layers:
        return list(range(num_layers))
    if keep <= 4:
        head = (keep + 1) // 2
        return list(range(head)) + list(range(num_layers - (keep - head), num_layers))
    inner = keep - 4
    span = num_layers - 4
    mid = sorted({2 + int(round((i + 0.5) * span / inner - 0.5)) for i in range(inner)})
    out = [0, 1] + mid + [num_layers - 2, num_layers - 1]
    i = 2
    while len(out) < keep:                       # rounding collisions: fill from the front of the middle
        if i not in out:
            out.append(i)
        i += 1
    return sorted(out)[:keep] if len(out) > keep else sorted(out)


def layer_skip_draft_of(engine, kept_layers):
    """A draft engine that runs only ``kept_layers`` of ``engine``'s layers (same weights, nothing copied) and reads /
    writes the cache rows of exactly those layers, so that it can share the target's KV cache like a full-depth draft
    (self-speculative decoding by layer skipping: Zhang et al., "Draft & Verify", 2023; the set of layers to skip is
    found per model by search on a few prompts there).  Combine with ``quantized_draft_of`` for a low-bit draft that
    also skips layers: ``layer_skip_draft_of(quantized_draft_of(target, Q.SYM_INT4), kept)``.  On an MI355X a full-depth
    int4 draft of a 7B model costs 1.64 ms next to a 2.8 ms fp16 target step, which no acceptance rate can turn into a
    gain worth having; a 12-of-32-layer int4 draft costs 0.65 ms."""
    from .decoder import NativeDecoder
    kept = [int(i) for i in kept_layers]
    n = len(engine.layers)
    assert kept and all(0 <= i < n for i in kept) and kept == sorted(set(kept)), "kept_layers: ascending layer indices"
    base = engine.cache_layer if engine.cache_layer is not None else list(range(n))
    draft = NativeDecoder(engine.cfg, engine.embed, [engine.layers[i] for i in kept], engine.final_norm, engine.lm_head,
                          device=engine.device, final_norm_b=engine.final_norm_b, pos_embed=engine.pos_embed,
                          embed_norm=engine.embed_norm)
    draft.cache_layer = [base[i] for i in kept]
    return draft


def search_kept_layers(target, full_draft, prompts, keep: int, max_new_tokens: int = 48, max_step_draft: int = 4,
                       candidates_per_round: Optional[int] = None, kv_fp8=False, log=None):
    """Which ``keep`` layers a layer-skipping draft of ``full_draft`` should run for ``target``: greedy backward
    elimination on calibration prompts (token id tensors [1, S]).  Starting from all layers, the layer whose removal
    costs the least acceptance is dropped, until ``keep`` remain; the acceptance of a candidate set is the share of
    its draft tokens that the target accepts while generating ``max_new_tokens`` per prompt with that draft (the
    selection criterion of Zhang et al., "Draft & Verify", with a greedy search in place of their Bayesian one).
    ``candidates_per_round`` limits how many removals are tried per round (evenly spread over the kept layers; the first
    and the last layer are never candidates).  Returns ``(kept_layers, acceptance)``.  One evaluation is one short
    generation per prompt: for a 32-layer model and 16 kept layers about 400 of them with every candidate tried."""
    n = len(full_draft.layers)
    kept = list(range(n))

    def acceptance(layers):
        draft = layer_skip_draft_of(full_draft, layers) if len(layers) < n else full_draft
        spec = SelfSpeculativeEngine(target, draft, max_step_draft=max_step_draft, adapt=False)
        drafted = matched = 0
        for ids in prompts:
            spec.generate(ids, max_new_tokens=max_new_tokens, kv_fp8=kv_fp8)
            drafted += spec.n_drafted
            matched += spec.n_matched
        return matched / max(1, drafted)

    score = acceptance(kept)
    while len(kept) > max(1, keep):
        pool = [i for i in kept if i not in (kept[0], kept[-1])] or kept[1:]
        if candidates_per_round and len(pool) > candidates_per_round:
            step = len(pool) / candidates_per_round
            pool = [pool[int(j * step)] for j in range(candidates_per_round)]
        best = None
        for cand in pool:
            trial = [i for i in kept if i != cand]
            sc = acceptance(trial)
            if best is None or sc > best[0]:
                best = (sc, cand, trial)
        score, dropped, kept = best
        if log is not None:
            log(f"dropped layer {dropped}: {len(kept)} layers, acceptance {score:.3f}")
    return kept, score
