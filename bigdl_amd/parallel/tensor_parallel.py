"""Tensor parallelism for low-bit models: Megatron-style column / row sharding of the quantized weights.

The reference obtains TP from DeepSpeed AutoTP (python/llm/example/GPU/Deepspeed-AutoTP/deepspeed_autotp.py:84-105:
``deepspeed.init_inference(mp_size=world)`` then ``optimize_model``; the all-reduce after row-parallel layers is the
``mp_group`` branch of LowBitLinear, python/llm/src/ipex_llm/transformers/low_bit_linear.py:712-722) and from vLLM's
parallel linear layers.  Here the sharding acts on the *quantized* buffers directly, so a rank never materializes
weights it does not own:

* host block layout: a row range is a contiguous byte range; a column range is a block range inside every row.
* T16 device layout: planes are [tile_row][k_tile][bytes]; a row range is one contiguous range per plane, a column
  range (multiple of 128) is a slice of the k_tile axis of every plane.

``tensor_parallel(model)`` shards a converted HuggingFace model in place: qkv / gate / up by output rows (whole
heads), o_proj / down_proj by input columns with ``mp_group`` set, so each decoder layer does two all-reduces of
[tokens, hidden].  ``shard_engine`` does the same for the native decoder engine.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn

from .. import qtypes as Q
from ..ops import host
from ..transformers.low_bit_linear import FP4Params, LowBitLinear, _HalfLinear


# ---------------------------------------------------------------------------------------------
# quantized buffer slicing
# ---------------------------------------------------------------------------------------------
def _host_rows(raw, qtype, n, k, r0, r1):
    row_bytes = raw.numel() // n
    return raw.view(n, row_bytes)[r0:r1].reshape(-1).contiguous()


def _host_cols(raw, qtype, n, k, c0, c1):
    qk, ts = host.ggml_qk_size(qtype), host.ggml_type_size(qtype)
    if c0 % qk or c1 % qk:
        raise ValueError(f"column range must be a multiple of the block size {qk}")
    return raw.view(n, k // qk, ts)[:, c0 // qk:c1 // qk].reshape(-1).contiguous()


def _t16_cols(buf, qtype, n, k, c0, c1):
    if c0 % 128 or c1 % 128:
        raise ValueError("column range of a tiled weight must be a multiple of 128")
    nt, kt = (n + 15) // 16, (k + 127) // 128
    flat = buf.view(torch.uint8).reshape(-1)
    parts, off = [], 0
    for sz in host.t16_plane_nbytes(qtype, n, k):
        if sz:
            parts.append(flat[off:off + sz].view(nt, kt, sz // (nt * kt))[:, c0 // 128:c1 // 128].reshape(-1))
        off += sz
    return torch.cat(parts).contiguous()


def slice_param(p: FP4Params, rows=None, cols=None) -> FP4Params:
    """A new quantized parameter holding rows [r0, r1) and/or columns [c0, c1) of ``p`` (same device / layout)."""
    n, k = p._shape
    data = p.data
    raw = data.view(torch.uint8) if data.dtype != torch.uint8 else data
    tiled = p.qlayout == "t16"
    if rows is not None:
        r0, r1 = rows
        raw = host.slice_t16_rows(raw, p.qtype, n, k, r0, r1 - r0) if tiled else _host_rows(raw.reshape(-1), p.qtype, n, k, r0, r1)
        n = r1 - r0
    if cols is not None:
        c0, c1 = cols
        raw = _t16_cols(raw, p.qtype, n, k, c0, c1) if tiled else _host_cols(raw.reshape(-1), p.qtype, n, k, c0, c1)
        k = c1 - c0
    if data.dtype != torch.uint8:
        raw = raw.view(data.dtype)
    return FP4Params(raw, requires_grad=False, quantized=True, _shape=(n, k), qtype=p.qtype, in_features=k,
                     host_only=p.host_only)


def _parts_ranges(sizes, rank, world, unit):
    """Row ranges that give this rank 1/world of every part of a merged projection (units = whole heads).  A part with
    fewer heads than ranks (the K / V heads of a grouped-query model at a high TP degree) is replicated instead:
    ``world / heads`` consecutive ranks keep the same head, which is the kv head their query heads belong to."""
    out, off = [], 0
    for s in sizes:
        heads = s // unit
        if heads < world and unit > 1:
            assert world % heads == 0, f"cannot place {heads} heads on {world} ranks"
            h = rank // (world // heads)
            out.append((off + h * unit, off + (h + 1) * unit))
        else:
            assert s % (unit * world) == 0, f"cannot split {heads} heads over {world} ranks"
            per = s // world
            out.append((off + rank * per, off + (rank + 1) * per))
        off += s
    return out


def shard_linear(lin: nn.Module, rank: int, world: int, mode: str, group=None, parts=None, unit=1):
    """In place.  mode "column": keep 1/world of the output rows (of each part in ``parts``); mode "row": keep
    1/world of the input columns and all-reduce the result over ``group``."""
    if isinstance(lin, LowBitLinear):
        w = lin.weight
        n, k = w._shape
        if mode == "column":
            ranges = _parts_ranges(parts or [n], rank, world, unit)
            pieces = [slice_param(w, rows=r) for r in ranges]
            if len(pieces) == 1:
                new = pieces[0]
            else:
                if pieces[0].qlayout == "t16":
                    raw = _concat_t16([q.data for q in pieces], w.qtype, [q._shape[0] for q in pieces], k)
                else:
                    raw = torch.cat([q.data.view(torch.uint8).reshape(-1) for q in pieces])
                tot = sum(q._shape[0] for q in pieces)
                new = FP4Params(raw.view(w.data.dtype) if w.data.dtype != torch.uint8 else raw, quantized=True,
                                _shape=(tot, k), qtype=w.qtype, in_features=k)
            lin.weight = new
            lin.out_len = lin.out_features = new._shape[0]
            if lin.bias is not None:
                lin.bias = nn.Parameter(torch.cat([lin.bias.data[a:b] for a, b in ranges]), requires_grad=False)
        else:
            assert k % world == 0
            per = k // world
            lin.weight = slice_param(w, cols=(rank * per, (rank + 1) * per))
            lin.in_len = lin.in_features = per
            lin.mp_group = group if group is not None else dist.group.WORLD
        lin.weight_shape = (lin.out_len, lin.in_len)
        return lin
    # dense layers (fp16 / bf16 / not converted)
    w = lin.weight.data
    n, k = w.shape
    if mode == "column":
        ranges = _parts_ranges(parts or [n], rank, world, unit)
        lin.weight = nn.Parameter(torch.cat([w[a:b] for a, b in ranges]).contiguous(), requires_grad=False)
        lin.out_features = lin.weight.shape[0]
        if hasattr(lin, "out_len"):
            lin.out_len = lin.out_features
        if lin.bias is not None:
            lin.bias = nn.Parameter(torch.cat([lin.bias.data[a:b] for a, b in ranges]), requires_grad=False)
    else:
        per = k // world
        lin.weight = nn.Parameter(w[:, rank * per:(rank + 1) * per].contiguous(), requires_grad=False)
        lin.in_features = per
        if hasattr(lin, "in_len"):
            lin.in_len = per
        if isinstance(lin, _HalfLinear):
            lin.mp_group = group if group is not None else dist.group.WORLD
        else:
            _wrap_row_parallel(lin, group)
    return lin


def _wrap_row_parallel(lin: nn.Linear, group):
    bias = lin.bias
    lin.bias = None

    def fwd(x, _orig=lin.forward.__func__ if hasattr(lin.forward, "__func__") else None):
        y = nn.functional.linear(x, lin.weight)
        dist.all_reduce(y, group=group)
        return y if bias is None else y + bias
    lin.forward = fwd


# ---------------------------------------------------------------------------------------------
# HuggingFace module path
# ---------------------------------------------------------------------------------------------
def tensor_parallel(model, group=None, rank=None, world=None):
    """Shard a converted (still on the host) decoder-only HF model over the ranks of ``group``."""
    if world is None:
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
    if world == 1:
        return model
    cfg = model.config
    hd = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
    for layer in model.model.layers:
        att = layer.self_attn
        if hasattr(att, "qkv_proj"):
            sizes = list(att._merged_sizes)
            shard_linear(att.qkv_proj, rank, world, "column", parts=sizes, unit=hd)
            att._merged_sizes = tuple(max(s // world, hd) for s in sizes)
        else:
            for name in ("q_proj", "k_proj", "v_proj"):
                shard_linear(getattr(att, name), rank, world, "column", unit=hd)
        shard_linear(att.o_proj, rank, world, "row", group=group)
        for a in ("num_heads", "num_key_value_heads"):
            if hasattr(att, a):
                setattr(att, a, max(getattr(att, a) // world, 1))
        if hasattr(att, "num_key_value_groups") and hasattr(att, "num_heads") and hasattr(att, "num_key_value_heads"):
            att.num_key_value_groups = att.num_heads // att.num_key_value_heads
        mlp = layer.mlp
        if all(hasattr(mlp, n) for n in ("gate_proj", "up_proj", "down_proj")) and not hasattr(mlp, "num_experts"):
            shard_linear(mlp.gate_proj, rank, world, "column")
            shard_linear(mlp.up_proj, rank, world, "column")
            shard_linear(mlp.down_proj, rank, world, "row", group=group)
    object.__setattr__(model, "tp_group", group)
    object.__setattr__(model, "tp_world", world)
    return model


# ---------------------------------------------------------------------------------------------
# native engine
# ---------------------------------------------------------------------------------------------
def _shard_lin(lin, rank, world, mode, parts=None, unit=1):
    from ..models.decoder import Lin
    if lin.fused:
        p = lin.param
        n, k = p._shape
        if mode == "column":
            ranges = _parts_ranges(parts or [n], rank, world, unit)
            bufs = [slice_param(p, rows=r) for r in ranges]
            raw = bufs[0].data
            if len(bufs) > 1:       # planes of the pieces are joined plane by plane
                raw = _concat_t16([b.data for b in bufs], p.qtype, [b._shape[0] for b in bufs], k)
            new = FP4Params(raw, quantized=True, _shape=(sum(b._shape[0] for b in bufs), k), qtype=p.qtype)
            bias = None if lin.bias is None else torch.cat([lin.bias[a:b] for a, b in ranges])
        else:
            per = k // world
            new = slice_param(p, cols=(rank * per, (rank + 1) * per))
            bias = lin.bias if rank == 0 else None      # added once, before the all-reduce sums the partials
        return Lin(new, bias)
    w = lin.w
    n, k = w.shape
    if mode == "column":
        ranges = _parts_ranges(parts or [n], rank, world, unit)
        bias = None if lin.bias is None else torch.cat([lin.bias[a:b] for a, b in ranges])
        return Lin(torch.cat([w[a:b] for a, b in ranges]).contiguous(), bias)
    per = k // world
    return Lin(w[:, rank * per:(rank + 1) * per].contiguous(), lin.bias if rank == 0 else None)


def _concat_t16(bufs, qtype, rows, k):
    """Row-concatenate tiled buffers (each a multiple of 16 rows): concatenate plane by plane."""
    planes = [[] for _ in range(4)]
    for b, n in zip(bufs, rows):
        assert n % 16 == 0
        flat, off = b.view(torch.uint8).reshape(-1), 0
        for i, sz in enumerate(host.t16_plane_nbytes(qtype, n, k)):
            if sz:
                planes[i].append(flat[off:off + sz])
            off += sz
    return torch.cat([x for pl in planes for x in pl]).contiguous()


def shard_engine(engine, rank: int, world: int, group=None, fast_allreduce=None):
    """Tensor-parallel view of a ``NativeDecoder``: attention heads and MLP columns are split over ``world`` ranks,
    the engine all-reduces after o_proj and down_proj (see ``NativeDecoder.tp_group``).  On GPUs the all-reduces go
    through the IPC kernels of ``parallel/xgmi_allreduce.py`` once their start-up self-check has passed on every rank
    (``fast_allreduce=False`` or ``BIGDL_XGMI_ALLREDUCE=0``: RCCL)."""
    import dataclasses
    if world == 1:
        return engine
    cfg = engine.cfg
    hd = cfg.head_dim
    assert cfg.num_heads % world == 0, "query heads must divide the TP degree"
    assert cfg.num_kv_heads % world == 0 or world % cfg.num_kv_heads == 0, "kv heads must divide, or be divided by, the TP degree"
    q, kv = cfg.q_size, cfg.kv_size
    for lw in engine.layers:
        if lw.mla is not None:
            # latent attention: the latents (``a`` and the norms) are shared by all heads and stay whole on every rank;
            # what expands them into heads is sharded by head -- q (or q_b) and both halves [K heads | V heads] of kv_b
            m = lw.mla
            if m.q is not None:
                m.q = _shard_lin(m.q, rank, world, "column", unit=hd)
            else:
                m.q_b = _shard_lin(m.q_b, rank, world, "column", unit=hd)
            m.kv_b = _shard_lin(m.kv_b, rank, world, "column", parts=[q, q], unit=hd)
        else:
            lw.qkv = _shard_lin(lw.qkv, rank, world, "column", parts=[q, kv, kv], unit=hd)
        lw.o = _shard_lin(lw.o, rank, world, "row")
        if lw.moe is not None:
            lw.moe.shard_experts(rank, world, group)
        else:
            if lw.gate is not None:        # (LayerNorm families have no gate projection)
                lw.gate = _shard_lin(lw.gate, rank, world, "column")
            lw.up = _shard_lin(lw.up, rank, world, "column")
            lw.down = _shard_lin(lw.down, rank, world, "row")
    engine.cfg = dataclasses.replace(cfg, num_heads=cfg.num_heads // world, num_kv_heads=max(cfg.num_kv_heads // world, 1),
                                     intermediate_size=cfg.intermediate_size // world)
    if getattr(engine, "alibi", None) is not None:
        per = cfg.num_heads // world
        engine.alibi = engine.alibi[rank * per:(rank + 1) * per].contiguous()
    engine.tp_group = group if group is not None else dist.group.WORLD
    engine.tp_world = world
    if fast_allreduce is None:
        import os
        fast_allreduce = os.environ.get("BIGDL_XGMI_ALLREDUCE", "1") != "0"
    if fast_allreduce and engine.device.type == "cuda":
        from .xgmi_allreduce import XgmiAllReduce
        engine.tp_fast_allreduce = XgmiAllReduce.create_verified(engine.tp_group, device=engine.device)
    engine.cache = None
    engine._graphs.clear()
    engine._static.clear()
    return engine
