"""Point-to-point transport between pipeline stages (used by the engine stages in parallel/pipeline.py and by the
serving worker in transformers/pipeline_parallel.py).

Hidden states travel forward on one communicator, sampled tokens travel from the last stage straight back to the first
on another: RCCL executes the operations of one communicator in issue order, so with a single communicator "send
micro-batch 1 forward" would queue behind "receive the token of micro-batch 0" and stall the ring.  The reference
moves the same tensors with ``dist.send`` / ``dist.recv`` plus a broadcast of every token
(python/llm/src/ipex_llm/transformers/pipeline_parallel.py:380-420, :760-800).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class PipelineLink:
    """Point-to-point transport of one stage.  ``ranks`` lists the global ranks of the stages in order (default: the
    whole world).  Over RCCL the tensors go GPU to GPU; over gloo (tests, two stages on one GPU) device tensors are
    staged through host memory."""

    def __init__(self, stage: Optional[int] = None, ranks: Optional[List[int]] = None):
        self.ranks = list(ranks) if ranks is not None else list(range(dist.get_world_size()))
        me = dist.get_rank()
        self.stage = self.ranks.index(me) if stage is None else stage
        self.n = len(self.ranks)
        self.first, self.last = self.stage == 0, self.stage == self.n - 1
        self.prev = self.ranks[(self.stage - 1) % self.n]
        self.next = self.ranks[(self.stage + 1) % self.n]
        self.head, self.tail = self.ranks[0], self.ranks[-1]
        self.host_staged = dist.get_backend() != "nccl"
        # two communicators: forward (hidden states) and back (tokens); every rank of the job takes part in new_group
        if self.n > 1:
            self.fwd = dist.new_group(self.ranks)
            self.back = dist.new_group(self.ranks)
        else:
            self.fwd = self.back = None

    # every call returns an object with .wait(); a buffer handed to send() must stay untouched until then
    def _send(self, t, dst, group):
        if self.host_staged and t.is_cuda:
            t = t.cpu()
        return dist.isend(t.contiguous(), dst=dst, group=group)

    def _recv(self, t, src, group):
        if self.host_staged and t.is_cuda:
            host = torch.empty(t.shape, dtype=t.dtype)
            dist.recv(host, src=src, group=group)
            t.copy_(host)
            return _Done()
        return dist.irecv(t, src=src, group=group)

    def send_hidden(self, t):
        return self._send(t, self.next, self.fwd)

    def recv_hidden(self, t):
        return self._recv(t, self.prev, self.fwd)

    def send_tokens(self, t):
        return self._send(t, self.head, self.back)

    def recv_tokens(self, t):
        return self._recv(t, self.tail, self.back)

    def broadcast_from_head(self, t):
        """``t`` of the first stage to every stage (same shape everywhere); returns the tensor on ``t``'s device."""
        if self.n == 1:
            return t
        x = t.cpu() if self.host_staged and t.is_cuda else t
        dist.broadcast(x, src=self.head, group=self.fwd)
        return x.to(t.device)


class _Done:
    def wait(self):
        return True
