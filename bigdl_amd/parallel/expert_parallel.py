"""Expert parallelism for mixture-of-experts decoders on the native engine.

Not present in the reference (its MoE blocks loop over all experts on one device:
python/llm/src/ipex_llm/transformers/models/qwen2_moe.py:376-452; pipeline parallelism only moves whole layers).  On
an 8 x MI355X node the experts of Mixtral / Qwen2-MoE are split over the ranks and attention is either

* replicated (``mode="a2a"``): every rank serves its own sequences (data-parallel attention, full KV bandwidth per
  rank) and only routed rows cross xGMI -- all-to-all dispatch / return in prefill, all-gather + reduce-scatter of
  the step's rows inside the decode graph;
* tensor-parallel (``mode="allreduce"``, what ``tensor_parallel.shard_engine`` sets up): tokens are replicated, each
  rank adds its experts' share and one all-reduce per MoE layer sums them.

Sizing for xGMI (7 links x ~153 GB/s per GPU, point to point): the a2a payload per layer is rows x hidden x 2 B split
over 7 peers, e.g. 1024 tokens x top-2 x 4096 x 2 B = 16 MiB out and back, ~2.3 MiB per link, all links busy at
once -- the reason to prefer it over a ring all-reduce of the full [tokens, hidden] block, which is bound by one link.
"""
from __future__ import annotations

import os

import torch.distributed as dist


def shard_engine_experts(engine, rank: int, world: int, group=None, mode: str = "a2a"):
    """Keep experts [rank*E/world, (rank+1)*E/world) of every MoE layer on this rank; attention, norms, embeddings and
    the lm_head stay replicated.  ``mode``: ``"a2a"`` (tokens sharded over ranks) or ``"allreduce"`` (tokens
    replicated)."""
    if world == 1:
        return engine
    group = group if group is not None else dist.group.WORLD
    n = 0
    for lw in engine.layers:
        if lw.moe is not None:
            lw.moe.shard_experts(rank, world, group, mode=mode)
            n += 1
    if n == 0:
        raise ValueError("the model has no mixture-of-experts layers to shard")
    if mode == "allreduce" and engine.device.type == "cuda" and os.environ.get("BIGDL_XGMI_ALLREDUCE", "1") != "0":
        # the per-layer sum of the experts' partial outputs goes through the IPC all-reduce kernels once their start-up
        # self-check has passed on every rank (parallel/xgmi_allreduce.py); otherwise the process group's all-reduce
        from .xgmi_allreduce import XgmiAllReduce
        fast = XgmiAllReduce.create_verified(group, device=engine.device)
        engine.tp_fast_allreduce = fast            # generate() checks its error word after the last step
        for lw in engine.layers:
            if lw.moe is not None:
                lw.moe.fast_allreduce = fast
    engine._graphs.clear()
    engine._static.clear()
    return engine
