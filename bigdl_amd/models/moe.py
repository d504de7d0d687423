"""Mixture-of-experts feed-forward for the native decoder engine (Mixtral, Qwen2-MoE layouts).

Decode step (graph-replayable, no host round trip): router GEMM -> ``moe_route`` kernel (softmax + top-k on the
device) -> routed gate/up GEMV with fused activation -> routed down GEMV scaled by the routing weight -> one
combine kernel (sum over the k slots + gated shared expert + residual).  Only the selected experts' weights are
streamed: for Mixtral-8x7B that is 2/8 of the expert bytes per token.

Prefill: tokens are sorted by expert on the device and the three expert products run as grouped dequant-fused MFMA
GEMMs over 64-row blocks (``gemm_lowbit_grouped``); dense (unquantized) experts run one GEMM per non-empty bucket.

Expert parallelism: ``expert_range=(lo, hi)`` keeps a contiguous slice of the experts on this rank.  Two layouts:

* tokens replicated (tensor-parallel attention next to it, ``ep_mode="allreduce"``): slots routed elsewhere produce zero
  rows and the partial outputs are summed with one all-reduce of [tokens, hidden];
* tokens sharded (every rank serves its own sequences with replicated attention, ``ep_mode="a2a"``): prefill
  dispatches each routed row to the rank that owns its expert with ``all_to_all_single`` (variable splits), runs
  the grouped GEMM on what arrived and returns the rows the same way; the graph-captured decode step uses the
  fixed-shape form of the same exchange (all-gather of the step's [batch, hidden] rows, local experts over all of them,
  reduce-scatter back), because a hipGraph cannot hold data-dependent split sizes.

Reference behaviour: python/llm/src/ipex_llm/transformers/models/qwen2_moe.py:376-452 (``qwen2moe_moeblock_forward``:
per-expert loops, ``xe_linear.get_moe_indexes`` on the host), models/phixtral.py:44-104; Mixtral and DeepSeek have no
forward of their own there (stock HuggingFace expert loops over ``LowBitLinear`` modules).
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn.functional as F

from .. import qtypes as Q


def grouped_moe_ffn(x, ids, weights, gate, up, down, qtype, n_experts, inter, hidden, act, zero_fill=False):
    """Routed expert FFN for many tokens on the grouped dequant-fused MFMA GEMM.

    x [T, H] fp16; ids int [T, k] (negative = expert not on this rank); weights [T, k]; gate / up / down: T16 buffers of
    the experts' matrices stacked by rows.  Rows are sorted by expert on the device and cut into 64-row blocks by a
    table kernel, so there is no host round trip (the reference buckets on the host:
    python/llm/src/ipex_llm/transformers/models/qwen2_moe.py:400-416).  Returns sum_k w * expert(x) as [T, H]."""
    from ..ops import hip
    t, k = ids.shape
    flat = ids.reshape(-1).long()
    key = torch.where(flat >= 0, flat, torch.full_like(flat, n_experts))      # non-local slots sort last
    order = torch.argsort(key, stable=True)
    tok = torch.div(order, k, rounding_mode="floor")
    rows, bm = t * k, 64
    counts = torch.bincount(key, minlength=n_experts + 1)[:n_experts].contiguous()
    table = hip.moe_block_table(counts, bm, (rows + bm - 1) // bm + n_experts)
    tok32 = tok.int()
    alloc = torch.zeros if zero_fill else torch.empty                            # non-local rows must stay zero
    g = alloc((rows, inter), dtype=torch.float16, device=x.device)
    hip.gemm_lowbit_grouped(x, gate, qtype, n_experts, inter, hidden, table, rows, bm, x_rows=tok32, out=g)
    u = hip.gemm_lowbit_grouped(x, up, qtype, n_experts, inter, hidden, table, rows, bm, x_rows=tok32)
    hip.act_mul(g, u, act)
    y = alloc((rows, hidden), dtype=torch.float16, device=x.device)
    hip.gemm_lowbit_grouped(g, down, qtype, n_experts, hidden, inter, table, rows, bm,
                            row_scale=weights.reshape(-1)[order].float().contiguous(), out=y)
    out = torch.zeros_like(x)
    out.index_add_(0, tok, y)
    return out


_MOE_SUM = os.environ.get("BIGDL_MOE_SUM", "0") == "1"
_FUSED_ROUTE = os.environ.get("BIGDL_MOE_FUSED_ROUTE", "1") != "0"     # 0: norm, router product and top-k as separate launches


class MoEWeights:
    def __init__(self, router, gate, up, down, qtype, num_experts, top_k, hidden, inter, act="silu", normalize=True,
                 shared=None, shared_router=None, expert_range=None, group=None):
        """gate/up: T16 buffers of [E*inter, hidden]; down: T16 buffer of [E*hidden, inter] (or dense fp16
        [E, inter, hidden] / [E, hidden, inter] tensors with qtype = FP16).  shared: (gate Lin, up Lin, down Lin)."""
        self.router = router.detach().to(torch.float16).contiguous()
        self.gate, self.up, self.down = gate, up, down
        self.qtype = qtype
        self.fused = qtype in Q.T16_QTYPES
        self.num_experts, self.top_k, self.hidden, self.inter = num_experts, top_k, hidden, inter
        self.act, self.normalize = act, normalize
        self.shared = shared
        self.shared_router = None if shared_router is None else shared_router.detach().to(torch.float16).contiguous()
        self.expert_range = expert_range        # (lo, hi) of the experts held locally, None = all
        self.group = group                      # process group for the expert-parallel all-reduce
        self.n_local = num_experts if expert_range is None else expert_range[1] - expert_range[0]
        self.ep_mode = "allreduce"              # "a2a": tokens are sharded over the expert-parallel ranks
        self.ep_capacity_factor = None          # > 0: fixed-shape prompt exchange (None: BIGDL_EP_CAPACITY_FACTOR)
        self.ep_dropped = 0                     # rows the fixed-shape exchange could not place (device counter)
        self.ep_rank, self.ep_world = 0, 1
        # group-limited routing (DeepSeek-V2): experts in n_group groups, only the topk_group best groups compete
        self.n_group, self.topk_group = 1, 1
        # DeepSeek-V3: sigmoid scores; experts are CHOSEN by score + route_bias (a learned load-balancing correction),
        # groups ranked by the sum of their two best; the weights are the unbiased scores of the chosen experts
        self.score, self.group_score, self.route_bias = "softmax", "max", None

    # ------------------------------------------------------------------ builders
    @classmethod
    def from_hf(cls, mlp, dc, lin_fn):
        """From a converted HF sparse block (router ``gate`` + ``LowBitExperts`` or dense fused experts)."""
        from ..transformers.low_bit_experts import LowBitExperts
        ex = mlp.experts
        router = mlp.gate.weight
        normalize = bool(getattr(mlp.gate, "norm_topk_prob", True))
        if isinstance(ex, LowBitExperts) and ex.gate_proj.weight.qlayout == "t16":
            gate, up, down, qt = ex.gate_proj.weight.data, ex.up_proj.weight.data, ex.down_proj.weight.data, ex.qtype
            act = ex._act_name() or "silu"
        elif isinstance(ex, LowBitExperts):
            from ..transformers.low_bit_linear import dequant_weight
            e, i, h = ex.num_experts, ex.intermediate_dim, ex.hidden_dim
            gate = dequant_weight(ex.gate_proj.weight, torch.float16).view(e, i, h)
            up = dequant_weight(ex.up_proj.weight, torch.float16).view(e, i, h)
            down = dequant_weight(ex.down_proj.weight, torch.float16).view(e, h, i)
            qt, act = Q.FP16, ex._act_name() or "silu"
        else:
            i = ex.gate_up_proj.shape[1] // 2
            gu = ex.gate_up_proj.data.to(torch.float16)
            gate, up, down = gu[:, :i].contiguous(), gu[:, i:].contiguous(), ex.down_proj.data.to(torch.float16)
            qt, act = Q.FP16, dc.act
        shared = shared_router = None
        if hasattr(mlp, "shared_expert"):
            se = mlp.shared_expert
            shared = (lin_fn(se.gate_proj), lin_fn(se.up_proj), lin_fn(se.down_proj))
            shared_router = lin_fn(mlp.shared_expert_gate).dense()      # [1, H]; may itself have been quantized
        return cls(router, gate, up, down, qt, dc.num_experts, dc.top_k, dc.hidden_size, dc.moe_intermediate_size,
                   act=act, normalize=normalize, shared=shared, shared_router=shared_router)

    def shard_experts(self, rank: int, world: int, group=None, mode: str = "allreduce"):
        """Keep experts [rank*E/world, (rank+1)*E/world) (expert parallelism).  T16 buffers are re-sliced plane by
        plane through the dense form once at load time."""
        assert self.num_experts % world == 0, "expert count must divide the expert-parallel degree"
        per = self.num_experts // world
        lo, hi = rank * per, (rank + 1) * per
        if self.fused:
            from ..ops.host import slice_t16_rows as requantize_rows
            self.gate = requantize_rows(self.gate, self.qtype, self.num_experts * self.inter, self.hidden, lo * self.inter, per * self.inter)
            self.up = requantize_rows(self.up, self.qtype, self.num_experts * self.inter, self.hidden, lo * self.inter, per * self.inter)
            self.down = requantize_rows(self.down, self.qtype, self.num_experts * self.hidden, self.inter, lo * self.hidden, per * self.hidden)
        else:
            self.gate, self.up, self.down = (t[lo:hi].contiguous() for t in (self.gate, self.up, self.down))
        self.expert_range, self.group, self.n_local = (lo, hi), group, per
        assert mode in ("allreduce", "a2a")
        self.ep_mode, self.ep_rank, self.ep_world = mode, rank, world
        return self

    # ------------------------------------------------------------------ pieces
    def route(self, x, local_only=True):
        """-> (expert ids int32 [T, k], weights fp32 [T, k]).  local_only: ids of experts on other ranks become -1
        and local ones are rebased to this rank's slice."""
        logits = F.linear(x, self.router.to(x.dtype))
        rng = self.expert_range if local_only else None
        if x.device.type == "cuda" and self.n_group <= 64 and self.num_experts % self.n_group == 0 \
                and self.num_experts <= 1024 and (self.n_group > 1 or self.score == "sigmoid"):
            from ..ops import hip
            bias = self.route_bias
            if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous()):
                bias = self.route_bias = bias.float().contiguous()
            return hip.moe_route(logits, self.top_k, self.normalize, expert_range=rng, sigmoid_scores=self.score == "sigmoid",
                                 route_bias=bias, n_group=self.n_group, topk_group=self.topk_group,
                                 group_top2=self.group_score == "top2sum")
        if self.n_group > 1 or self.score == "sigmoid":
            # scores over all experts, experts outside the topk_group best groups drop out, weights are the scores of the
            # chosen experts (the DeepSeek-V2 / V3 models' own ``MoEGate``); plain tensor ops, replayable inside the
            # decode graph
            p = torch.sigmoid(logits.float()) if self.score == "sigmoid" else torch.softmax(logits.float(), -1)
            choice = p if self.route_bias is None else p + self.route_bias.float()
            if self.n_group > 1:
                grouped = choice.view(choice.shape[0], self.n_group, -1)
                gs = grouped.topk(2, dim=-1).values.sum(-1) if self.group_score == "top2sum" else grouped.amax(-1)
                keep = torch.zeros_like(gs).scatter_(1, gs.topk(self.topk_group, dim=-1).indices, 1.0)
                # (masked to 0, not -inf, as the model's own router does)
                choice = choice.masked_fill(keep.repeat_interleave(self.num_experts // self.n_group, 1) == 0, 0.0)
            ids = torch.topk(choice, self.top_k, dim=-1).indices
            w = p.gather(1, ids)
            if self.normalize:
                w = w / (w.sum(-1, keepdim=True) + 1e-20)
            ids = ids.int()
            if rng is not None:
                inside = (ids >= rng[0]) & (ids < rng[1])
                ids = torch.where(inside, ids - rng[0], torch.full_like(ids, -1))
            return ids.contiguous(), w.contiguous()
        if x.device.type == "cuda":
            from ..ops import hip
            return hip.moe_route(logits, self.top_k, self.normalize, expert_range=rng)
        p = torch.softmax(logits.float(), -1)
        w, ids = torch.topk(p, self.top_k, dim=-1)
        if self.normalize:
            w = w / w.sum(-1, keepdim=True)
        ids = ids.int()
        if rng is not None:
            keep = (ids >= rng[0]) & (ids < rng[1])
            ids = torch.where(keep, ids - rng[0], torch.full_like(ids, -1))
        return ids, w

    def _expert_w(self, which, e, dtype=torch.float16):
        from ..ops import hip
        buf = getattr(self, which)
        n, k = (self.hidden, self.inter) if which == "down" else (self.inter, self.hidden)
        if self.fused:
            return hip.dequant_t16(buf, self.qtype, n, k, dtype=dtype, total_rows=self.n_local * n, row0=e * n)
        return buf[e]

    def _shared_out(self, engine, x):
        """(shared expert output [T, H], gate logits [T, 1]) or (None, None)"""
        if self.shared is None:
            return None, None
        from ..ops import hip
        sg, su, sd = self.shared
        t = x.shape[0]
        if t <= 16 and sg.fused and su.fused and sd.fused:
            a = hip.gemv_lowbit(x, sg.w, sg.qtype, sg.n, sg.k, w2=su.w, act=self.act)
            y = hip.gemv_lowbit(a, sd.w, sd.qtype, sd.n, sd.k)
        elif t <= 256 and sg.fused and su.fused and sd.fused:
            a = hip.act_mul(hip.gemm_lowbit(x, sg.w, sg.qtype, sg.n, sg.k), hip.gemm_lowbit(x, su.w, su.qtype, su.n, su.k),
                            self.act)
            y = hip.gemm_lowbit(a, sd.w, sd.qtype, sd.n, sd.k)
        else:
            a = hip.act_mul(F.linear(x, sg.dense()), F.linear(x, su.dense()), self.act)
            y = F.linear(a, sd.dense())
        gl = F.linear(x, self.shared_router) if self.shared_router is not None else None
        return y, gl

    def _reduce(self, t):
        if self.group is not None or (self.expert_range is not None and torch.distributed.is_initialized()):
            fast = getattr(self, "fast_allreduce", None)
            if fast is not None and fast.usable(t):
                fast(t)                   # IPC kernel: one launch, capturable in the decode graph
            else:
                torch.distributed.all_reduce(t, group=self.group)
        return t

    # ------------------------------------------------------------------ decode
    def forward_decode(self, engine, h, gamma):
        """h [m, H] fp16 (residual stream), updated in place: h += moe(rmsnorm(h))"""
        from ..ops import hip
        cfg = engine.cfg
        a2a = self.ep_mode == "a2a" and self.ep_world > 1
        groups_ok = self.n_group <= 64 and self.num_experts % self.n_group == 0
        if _FUSED_ROUTE and not a2a and groups_ok and h.shape[0] <= 256 and hip.moe_norm_route_ok(h, gamma, self.router, self.num_experts):
            # norm + router product + (group-limited) top-k in one launch: three launches otherwise, a dozen with the
            # DeepSeek routers' selection written as tensor ops, each mostly fixed cost at a few rows
            bias = self.route_bias
            if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous()):
                bias = self.route_bias = bias.float().contiguous()
            x, ids, w = hip.moe_norm_route(h, gamma, cfg.rms_eps, self.router, self.top_k, self.normalize,
                                           expert_range=self.expert_range, offset=cfg.norm_offset,
                                           sigmoid_scores=self.score == "sigmoid", route_bias=bias, n_group=self.n_group,
                                           topk_group=self.topk_group, group_top2=self.group_score == "top2sum")
        else:
            x = hip.rms_norm(h, gamma, cfg.rms_eps, offset=cfg.norm_offset)
            if a2a:
                y = self.forward_gather_scatter(x)
                sy, gl = self._shared_out(engine, x)
                hip.moe_combine(y, 1, h, shared=sy, shared_gate=gl, residual=h)
                return h
            ids, w = self.route(x)
        m, k = x.shape[0], self.top_k
        if self.fused:
            if _MOE_SUM and self.shared is None and self.expert_range is None and m <= 16 and h.dtype == torch.float16 \
                    and hip.gemv_lowbit_moe_sum_ok(k, self.inter) and self.hidden % 16 == 0:
                # routing weights applied to act(g) * u, then the k down products of a token as one product over the
                # concatenated reduction axis with the residual add in its epilogue: two launches, no combine.
                # BIGDL_MOE_SUM=1 only: on Mixtral-8x7B (one MI355X, batch 1) 2.60 ms per token against 2.51 with the
                # separate combine launch: 256 workgroups that walk two experts each stream slower than 512 that walk
                # one, by more than the 4.6 us launch they save.
                a = hip.gemv_lowbit_moe(x, self.gate, self.qtype, self.n_local, self.inter, self.hidden, ids.view(-1),
                                        w2=self.up, slot_xdiv=k, act=self.act, slot_scale=w.view(-1))
                return hip.gemv_lowbit_moe_sum(a, self.down, self.qtype, self.n_local, self.hidden, self.inter,
                                               ids.view(-1), k, residual=h, out=h)
            a = hip.gemv_lowbit_moe(x, self.gate, self.qtype, self.n_local, self.inter, self.hidden, ids.view(-1),
                                    w2=self.up, slot_xdiv=k, act=self.act)
            y = hip.gemv_lowbit_moe(a, self.down, self.qtype, self.n_local, self.hidden, self.inter, ids.view(-1),
                                    slot_scale=w.view(-1))
        else:
            y = self._dense_slots(x, ids, w)
        sy, gl = self._shared_out(engine, x)
        if self.expert_range is not None:
            part = torch.empty_like(h)
            hip.moe_combine(y, k, part)
            self._reduce(part)
            y, k = part, 1
        hip.moe_combine(y, k, h, shared=sy, shared_gate=gl, residual=h)
        return h

    def _dense_slots(self, x, ids, w):
        """Unfused fallback for decode with dense expert weights: gather the selected experts and bmm."""
        m, k = ids.shape
        sel = ids.view(-1).clamp_min(0).long()
        xs = x.repeat_interleave(k, 0).unsqueeze(1)                                   # [m*k, 1, H]
        g = torch.bmm(xs, self.gate[sel].transpose(1, 2))
        u = torch.bmm(xs, self.up[sel].transpose(1, 2))
        from ..ops import hip
        a = hip.act_mul(g.squeeze(1).contiguous(), u.squeeze(1).contiguous(), self.act)
        y = torch.bmm(a.unsqueeze(1), self.down[sel].transpose(1, 2)).squeeze(1)
        scale = (w.view(-1) * (ids.view(-1) >= 0)).to(y.dtype)
        return (y * scale[:, None]).contiguous()

    # ------------------------------------------------------------------ expert FFN over rows with local expert ids
    def local_ffn(self, x, ids, w):
        """sum_k w[t, k] * expert_{ids[t, k]}(x[t]) over this rank's experts (ids < 0 contribute nothing) -> [T, H]"""
        t, k = ids.shape
        if self.fused and x.device.type == "cuda":
            return grouped_moe_ffn(x, ids, w, self.gate, self.up, self.down, self.qtype, self.n_local, self.inter,
                                   self.hidden, self.act, zero_fill=True)
        from ..models.arch import _ACTS
        act = _ACTS[self.act]
        out = torch.zeros_like(x)
        flat = ids.reshape(-1)
        wf = w.reshape(-1).to(x.dtype)
        tok = torch.arange(t, device=x.device).repeat_interleave(k)
        for e in range(self.n_local):
            sel = (flat == e).nonzero().squeeze(-1)
            if sel.numel() == 0:
                continue
            xe = x[tok[sel]]
            g, u, d = (self._expert_w(n, e, x.dtype) for n in ("gate", "up", "down"))
            out.index_add_(0, tok[sel], F.linear(act(F.linear(xe, g)) * F.linear(xe, u), d) * wf[sel, None])
        return out

    def forward_a2a(self, x):
        """Tokens sharded over the expert-parallel group: dispatch routed rows to the owners of their experts,
        compute there, return them.  Three collectives per layer: the split sizes, one dispatch whose rows carry
        their local expert id in 8 spare halves behind the hidden vector, one return.

        The split sizes are read on the host once per layer, because ``all_to_all_single`` takes them as host lists.
        The fixed-shape alternative (every rank sends a capacity-padded block to every peer) would either move
        world/capacity_factor times the bytes over links that are the bottleneck of this exchange (xGMI is
        point-to-point: an all-to-all is bound per link) or drop rows that overflow; at prompt sizes one host read
        (~50 us) per layer is the cheaper of the two by default.  ``forward_a2a_capacity`` is that alternative, for
        callers that want the prompt pass free of host reads.  Decode steps use ``forward_gather_scatter`` (fixed
        shape, graph-captured)."""
        import torch.distributed as dist
        cap = float(os.environ.get("BIGDL_EP_CAPACITY_FACTOR", "0") or 0) if self.ep_capacity_factor is None else self.ep_capacity_factor
        if cap > 0:
            return self.forward_a2a_capacity(x, cap)
        t, k = x.shape[0], self.top_k
        hdim = x.shape[1]
        per = self.n_local
        ids, w = self.route(x, local_only=False)                     # global expert ids
        flat = ids.reshape(-1).long()
        dest = torch.div(flat, per, rounding_mode="floor")
        order = torch.argsort(dest, stable=True)
        tok = torch.div(order, k, rounding_mode="floor")
        send_counts = torch.bincount(dest, minlength=self.ep_world)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        both = torch.stack([send_counts, recv_counts]).tolist()     # the one host round trip: split sizes
        send_n, recv_n = both
        # payload rows: [hidden | local expert id as the raw bits of an int32 | padding to 16 bytes]
        pad = 8 if x.element_size() == 2 else 4
        send = x.new_zeros((tok.numel(), hdim + pad))
        send[:, :hdim] = x[tok]
        eid = (flat[order] % per).to(torch.int32)
        send[:, hdim:hdim + 4 // x.element_size()] = eid.view(-1, 1).view(x.dtype)
        recv = x.new_empty((sum(recv_n), hdim + pad))
        dist.all_to_all_single(recv, send, recv_n, send_n, group=self.group)
        if recv.shape[0]:
            recv_e = recv[:, hdim:hdim + 4 // x.element_size()].contiguous().view(torch.int32).view(-1)
            y = self.local_ffn(recv[:, :hdim].contiguous(), recv_e.view(-1, 1), torch.ones(recv_e.numel(), 1, device=x.device))
        else:
            y = recv[:, :hdim]
        back = x.new_empty((tok.numel(), hdim))
        dist.all_to_all_single(back, y.contiguous(), send_n, recv_n, group=self.group)
        out = torch.zeros_like(x)
        out.index_add_(0, tok, back * w.reshape(-1)[order].to(x.dtype)[:, None])
        return out

    def forward_a2a_capacity(self, x, factor: float):
        """The same exchange with no host read: every rank sends a block of ``capacity = ceil(factor * T * top_k /
        world)`` rows to every peer (unused rows carry expert id -1 and cost no FFN work), so both all-to-alls have
        equal, known splits and the whole layer stays on the device (it can sit inside a captured graph).  A rank
        that routes more than ``capacity`` rows to one peer drops the excess (they contribute zero, as in
        capacity-factor MoE training); ``factor = world`` can never drop.  ``self.ep_dropped`` counts dropped rows on
        the device.  Selected by ``BIGDL_EP_CAPACITY_FACTOR`` or ``ep_capacity_factor``; bytes over the links grow by
        ``factor`` over the balanced case."""
        import torch.distributed as dist
        t, k, W = x.shape[0], self.top_k, self.ep_world
        hdim, per = x.shape[1], self.n_local
        n = t * k
        C = min(n, max(1, math.ceil(factor * n / W)))
        ids, w = self.route(x, local_only=False)
        flat = ids.reshape(-1).long()
        dest = torch.div(flat, per, rounding_mode="floor")
        order = torch.argsort(dest, stable=True)
        tok = torch.div(order, k, rounding_mode="floor")
        dsort = dest[order]
        counts = torch.zeros(W, dtype=torch.long, device=x.device).scatter_add_(0, dest, torch.ones_like(dest))
        start = counts.cumsum(0) - counts
        pos = torch.arange(n, device=x.device) - start[dsort]
        keep = pos < C
        slot = torch.where(keep, dsort * C + pos, torch.full_like(pos, W * C))        # row W*C: where the excess lands
        pad = 8 if x.element_size() == 2 else 4
        ew = 4 // x.element_size()
        send = x.new_zeros((W * C + 1, hdim + pad))
        send[:, hdim:hdim + ew] = torch.full((1,), -1, dtype=torch.int32, device=x.device).view(x.dtype)
        rows = x.new_empty((n, hdim + pad))
        rows[:, :hdim] = x[tok]
        rows[:, hdim:hdim + ew] = (flat[order] % per).to(torch.int32).view(-1, 1).view(x.dtype)
        rows[:, hdim + ew:] = 0
        send.index_copy_(0, slot, rows)
        recv = x.new_empty((W * C, hdim + pad))
        dist.all_to_all_single(recv, send[:W * C], group=self.group)
        recv_e = recv[:, hdim:hdim + ew].contiguous().view(torch.int32).view(-1, 1)
        y = self.local_ffn(recv[:, :hdim].contiguous(), recv_e, torch.ones(W * C, 1, device=x.device))
        back = x.new_empty((W * C + 1, hdim))
        dist.all_to_all_single(back[:W * C], y.contiguous(), group=self.group)
        back[W * C] = 0
        self.ep_dropped = self.ep_dropped + (~keep).sum()
        out = torch.zeros_like(x)
        out.index_add_(0, tok, back[slot] * w.reshape(-1)[order].to(x.dtype)[:, None])
        return out

    def forward_gather_scatter(self, x):
        """Fixed-shape form of the exchange for the graph-captured decode step: every rank contributes the same number
        of rows; all-gather them, run the local experts over all rows, reduce-scatter the partial sums back."""
        import torch.distributed as dist
        allx = x.new_empty((x.shape[0] * self.ep_world, x.shape[1]))
        dist.all_gather_into_tensor(allx, x.contiguous(), group=self.group)
        ids, w = self.route(allx, local_only=True)
        part = self.local_ffn(allx, ids, w)
        out = torch.empty_like(x)
        dist.reduce_scatter_tensor(out, part.contiguous(), group=self.group)
        return out

    # ------------------------------------------------------------------ prefill
    def forward_dense(self, x, engine=None):
        """x [T, H] fp16 (already normalized) -> moe(x) [T, H]"""
        from ..ops import hip
        t, k = x.shape[0], self.top_k
        if self.ep_mode == "a2a" and self.ep_world > 1:
            out = self.forward_a2a(x)
            sy, gl = self._shared_out(engine, x)
            if sy is not None:
                out = out + (torch.sigmoid(gl.float()).to(sy.dtype) * sy if gl is not None else sy)
            return out
        ids, w = self.route(x)
        if self.fused and t * k <= 64:
            a = hip.gemv_lowbit_moe(x, self.gate, self.qtype, self.n_local, self.inter, self.hidden, ids.view(-1),
                                    w2=self.up, slot_xdiv=k, act=self.act)
            y = hip.gemv_lowbit_moe(a, self.down, self.qtype, self.n_local, self.hidden, self.inter, ids.view(-1),
                                    slot_scale=w.view(-1))
            out = y.view(t, k, -1).sum(1)
        else:
            if self.fused:
                out = grouped_moe_ffn(x, ids, w, self.gate, self.up, self.down, self.qtype, self.n_local, self.inter,
                                      self.hidden, self.act, zero_fill=self.expert_range is not None)
            else:
                flat = ids.view(-1).long()
                key = torch.where(flat >= 0, flat, torch.full_like(flat, self.n_local))
                order = torch.argsort(key, stable=True)
                tok = torch.div(order, k, rounding_mode="floor")
                counts = torch.bincount(key, minlength=self.n_local + 1).tolist()
                xs = x[tok]
                ws = w.view(-1)[order].to(x.dtype)
                out = torch.zeros_like(x)
                off = 0
                for e in range(self.n_local):
                    c = counts[e]
                    if c == 0:
                        continue
                    xe = xs[off:off + c]
                    a = hip.act_mul(F.linear(xe, self._expert_w("gate", e)), F.linear(xe, self._expert_w("up", e)), self.act)
                    ye = F.linear(a, self._expert_w("down", e))
                    out.index_add_(0, tok[off:off + c], ye * ws[off:off + c, None])
                    off += c
        if self.expert_range is not None:
            self._reduce(out)
        sy, gl = self._shared_out(engine, x)
        if sy is not None:
            out = out + (torch.sigmoid(gl.float()).to(sy.dtype) * sy if gl is not None else sy)
        return out

    @property
    def nbytes(self):
        tot = sum(t.numel() * t.element_size() for t in (self.gate, self.up, self.down, self.router))
        if self.shared is not None:
            tot += sum(l.nbytes for l in self.shared)
        return tot

    def active_bytes(self):
        """Bytes streamed per decoded token (top_k experts + shared expert + router)."""
        per = sum(t.numel() * t.element_size() for t in (self.gate, self.up, self.down)) // max(1, self.n_local)
        tot = per * self.top_k + self.router.numel() * 2
        if self.shared is not None:
            tot += sum(l.nbytes for l in self.shared)
        return tot
