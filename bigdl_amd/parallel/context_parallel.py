"""Context (sequence) parallel prefill for very long prompts: the prompt is cut into 2W chunks, rank r takes chunks r
and 2W-1-r (so every rank gets the same amount of causal attention work), every layer's new K / V chunks are
all-gathered into every rank's KV cache, and each rank runs flash attention for its own queries over the keys before
them.  The linear layers, norms and the MLP see only S / W tokens per rank.

The reference has no sequence parallelism: it bounds long prompts on one device with fp8 KV, SnapKV compression and
last-token lm_head (python/llm/src/ipex_llm/transformers/kv.py:171-243, models/utils.py:77-99, 317-330;
example/GPU/Long-Context/README.md:13-31, at most ~12 K tokens on a 16 GB card).  On 8 x 288 GB the memory motive is
gone; what remains is first-token latency, which this divides by W for the linear part and by about W for attention.

Why all-gather of K / V and not a ring: xGMI is fully connected, so an all-gather keeps all seven links of every
GPU busy for one step, where a ring would serialize W - 1 neighbour hops on two links; K / V of one layer (fp8 cache:
2 * S * Hkv * D bytes) are small next to the attention FLOPs they feed (S^2), and after the gather every rank holds
the complete cache, so decode can go on replicated (or tensor parallel) without a second redistribution.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def zigzag_chunks(world: int, rank: int) -> Tuple[int, int]:
    """chunk indices (of 2 * world equal chunks) owned by ``rank``: one early, one late"""
    return rank, 2 * world - 1 - rank


def chunk_owner_order(world: int) -> List[Tuple[int, int]]:
    """for every rank in order, its two chunk indices: the layout of an all-gathered buffer"""
    return [zigzag_chunks(world, r) for r in range(world)]


def local_token_index(seq_len: int, world: int, rank: int, device=None) -> torch.Tensor:
    """positions of the prompt tokens this rank processes (its two chunks, in order)"""
    assert seq_len % (2 * world) == 0, "prompt length must be a multiple of 2 * world (pad on the left)"
    n = seq_len // (2 * world)
    a, b = zigzag_chunks(world, rank)
    return torch.cat([torch.arange(a * n, (a + 1) * n, device=device), torch.arange(b * n, (b + 1) * n, device=device)])


def _world(group) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _rank(group) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def gather_kv_chunks(k_cache: torch.Tensor, v_cache: torch.Tensor, chunk_len: int, group=None) -> None:
    """k_cache [B, Hkv, Lmax, D] and v_cache = V^T [B, Hkv, D, Lmax] (any dtype: fp16 or e5m2 bytes) hold this rank's
    two chunks at their global positions; on return they hold every rank's chunks, i.e. positions
    [0, 2 * world * chunk_len).  One all-gather for K and V together."""
    world, rank = _world(group), _rank(group)
    if world == 1:
        return
    n = chunk_len
    b, hkv, _, d = k_cache.shape
    mine = zigzag_chunks(world, rank)
    # pack: [2 chunks][K | V^T] as flat bytes of one dtype
    parts = []
    for c in mine:
        parts.append(k_cache[:, :, c * n:(c + 1) * n, :].reshape(-1))
        parts.append(v_cache[:, :, :, c * n:(c + 1) * n].reshape(-1))
    send = torch.cat(parts)
    recv = torch.empty(world * send.numel(), dtype=send.dtype, device=send.device)
    try:
        dist.all_gather_into_tensor(recv, send, group=group)
    except (RuntimeError, NotImplementedError):
        dist.all_gather(list(recv.chunk(world)), send, group=group)
    per = b * hkv * n * d
    for r, chunks in enumerate(chunk_owner_order(world)):
        if r == rank:
            continue
        base = r * send.numel()
        for j, c in enumerate(chunks):
            kk = recv[base + (2 * j) * per: base + (2 * j + 1) * per].view(b, hkv, n, d)
            vv = recv[base + (2 * j + 1) * per: base + (2 * j + 2) * per].view(b, hkv, d, n)
            k_cache[:, :, c * n:(c + 1) * n, :].copy_(kk)
            v_cache[:, :, :, c * n:(c + 1) * n].copy_(vv)


def broadcast_from_last_token_owner(t: torch.Tensor, group=None) -> torch.Tensor:
    """the last prompt token lives in chunk 2W-1, which rank 0 owns: its logits go to everyone"""
    if _world(group) > 1:
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(t, src=src, group=group)
    return t
