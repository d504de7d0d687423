"""Batched tensor-parallel serving: ``shard_engine`` + ``ContinuousBatcher`` on every rank, one schedule.

The reference's tensor-parallel server batches waiting requests on rank 0 (``max_num_seqs`` /
``max_num_batched_tokens``) and ships each batch to the other ranks as pickled Python objects
(python/llm/example/GPU/Deepspeed-AutoTP-FastAPI/serving.py:335-411: ``dist.broadcast_object_list``).  Here every rank
owns a shard of the native engine and an identical ``ContinuousBatcher``; all that has to travel is *what rank 0
admitted or aborted this step*.  That schedule is two fixed-shape int64 tensors: a 4-word header every step, and --
only on steps that admit or abort something -- a payload block of request records and token ids.  No pickling, no
shape negotiation, and because the batchers are deterministic (same requests in the same order, identical logits
after the all-reduce, identically seeded samplers) every rank takes the same scheduling decisions, captures the same
graphs and enters the same collectives without any further coordination.

The schedule travels on a host-side (gloo) control group: followers block in a host receive while idle instead of
parking a collective kernel on the GPU, and the RCCL communicator carries nothing but the engine's all-reduces.
"""
from __future__ import annotations

import collections
import struct
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from .continuous import ContinuousBatcher, SeqRequest

HEADER_WORDS = 4                      # [new requests, aborted requests, prompt tokens in the payload, stop]
META_WORDS = 14                       # per admitted request, see _pack_request
MAX_EOS = 4


def _f2i(x: float) -> int:
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


def _i2f(x: int) -> float:
    return struct.unpack("<d", struct.pack("<q", int(x)))[0]


class TensorParallelBatcher:
    """One instance per rank.  Rank ``src`` calls ``submit`` / ``abort`` and drives ``step``; the other ranks call
    ``follow`` (or ``step`` in lock-step).  ``step`` returns the batcher's {request_id: token} on every rank."""

    def __init__(self, engine, group=None, control_group=None, src: int = 0, max_admit_per_step: int = 8,
                 max_batched_tokens: int = 8192, seed: int = 0, **batcher_kw):
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.src, self.is_src = src, self.rank == src
        self.max_admit, self.max_tokens, self.max_abort = max_admit_per_step, max_batched_tokens, 64
        if control_group is None and self.world > 1:
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            control_group = group if dist.get_backend(group) == "gloo" else dist.new_group(ranks, backend="gloo")
        self.control = control_group
        self.src_global = dist.get_global_rank(group, src) if (group is not None and self.world > 1) else src
        torch.manual_seed(seed)           # samplers draw the same numbers on every rank
        if torch.cuda.is_available():
            torch.cuda.manual_seed(seed)
        self.batcher = ContinuousBatcher(engine, **batcher_kw)
        self.requests: Dict[int, SeqRequest] = {}          # serial -> request (every rank)
        self._by_id: Dict[str, int] = {}
        self._pending = collections.deque()                # src: submitted, not yet broadcast
        self._aborts: List[int] = []
        self._serial = 0
        self._header = torch.zeros(HEADER_WORDS, dtype=torch.int64)
        self._payload = torch.zeros(self.max_admit * META_WORDS + self.max_abort + self.max_tokens, dtype=torch.int64)
        self.stopped = False
        self.steps = 0
        self.payload_steps = 0
        self.heartbeat_s = 5.0 if self.world > 1 else None    # idle source: an empty step this often (followers' receive)

    # ------------------------------------------------------------------ rank src: queue
    def submit(self, req: SeqRequest) -> SeqRequest:
        assert self.is_src, "requests enter on the source rank"
        if len(req.input_ids) > self.max_tokens:
            raise ValueError(f"request {req.request_id}: prompt longer than max_batched_tokens ({self.max_tokens})")
        if len(req.input_ids) == 0 or len(req.input_ids) + req.max_new_tokens > self.batcher.cache.capacity:
            raise ValueError(f"request {req.request_id}: prompt + max_new_tokens does not fit a cache slot")
        if isinstance(req.eos_token_id, int):
            req.eos_token_id = [req.eos_token_id]
        if req.eos_token_id and len(req.eos_token_id) > MAX_EOS:
            raise ValueError(f"at most {MAX_EOS} end-of-sequence ids per request")
        self._pending.append(req)
        return req

    def abort(self, request_id: str):
        assert self.is_src
        for req in list(self._pending):
            if req.request_id == request_id:
                self._pending.remove(req)
                req.finished, req.finish_reason = True, "abort"
                return
        serial = self._by_id.get(request_id)
        if serial is not None:
            self._aborts.append(serial)

    def has_work(self) -> bool:
        return bool(self._pending or self._aborts or self.batcher.has_work())

    # ------------------------------------------------------------------ the schedule
    def _pack_request(self, req: SeqRequest, serial: int, at: int):
        eos = list(req.eos_token_id or [])
        rec = [serial, len(req.input_ids), req.max_new_tokens, req.min_new_tokens, int(req.do_sample),
               _f2i(req.temperature), int(req.top_k), _f2i(req.top_p), len(eos)] + eos + [0] * (MAX_EOS - len(eos))
        rec += [0] * (META_WORDS - len(rec))
        self._payload[at:at + META_WORDS] = torch.tensor(rec, dtype=torch.int64)

    def _publish(self, stop: bool):
        """src: decide what enters this step and write header + payload."""
        new, ntok = [], 0
        while self._pending and len(new) < self.max_admit and ntok + len(self._pending[0].input_ids) <= self.max_tokens:
            req = self._pending.popleft()
            self._serial += 1
            new.append((self._serial, req))
            ntok += len(req.input_ids)
        aborts, self._aborts = self._aborts[:self.max_abort], self._aborts[self.max_abort:]
        self._header[:] = torch.tensor([len(new), len(aborts), ntok, int(stop)])
        tok_at = self.max_admit * META_WORDS + self.max_abort
        for j, (serial, req) in enumerate(new):
            self._pack_request(req, serial, j * META_WORDS)
            self._payload[tok_at:tok_at + len(req.input_ids)] = torch.tensor(req.input_ids, dtype=torch.int64)
            tok_at += len(req.input_ids)
        if aborts:
            a0 = self.max_admit * META_WORDS
            self._payload[a0:a0 + len(aborts)] = torch.tensor(aborts, dtype=torch.int64)
        return new

    def _exchange(self, stop: bool = False):
        """Every rank: after this call the header and (if announced) the payload hold the source's schedule.  Returns
        the new requests as (serial, SeqRequest) and the aborted serials."""
        mine = self._publish(stop) if self.is_src else None
        if self.world > 1:
            dist.broadcast(self._header, src=self.src_global, group=self.control)
        n_new, n_abort, ntok, stop_flag = self._header.tolist()
        if stop_flag:
            self.stopped = True
        if (n_new or n_abort) and self.world > 1:
            dist.broadcast(self._payload, src=self.src_global, group=self.control)
        self.payload_steps += bool(n_new or n_abort)
        a0 = self.max_admit * META_WORDS
        aborts = self._payload[a0:a0 + n_abort].tolist()
        if self.is_src:
            return mine, aborts
        new, tok_at = [], a0 + self.max_abort
        for j in range(n_new):
            rec = self._payload[j * META_WORDS:(j + 1) * META_WORDS].tolist()
            serial, plen, max_new, min_new, do_sample, temp, top_k, top_p, n_eos = rec[:9]
            req = SeqRequest(request_id=f"tp-{serial}", input_ids=self._payload[tok_at:tok_at + plen].tolist(),
                             max_new_tokens=max_new, min_new_tokens=min_new, do_sample=bool(do_sample),
                             temperature=_i2f(temp), top_k=top_k, top_p=_i2f(top_p),
                             eos_token_id=rec[9:9 + n_eos] or None)
            tok_at += plen
            new.append((serial, req))
        return new, aborts

    # ------------------------------------------------------------------ one iteration, every rank
    def step(self, stop: bool = False) -> Optional[Dict[str, int]]:
        new, aborts = self._exchange(stop)
        for serial, req in new:
            self.requests[serial] = req
            self._by_id[req.request_id] = serial
            self.batcher.submit(req)
        for serial in aborts:
            req = self.requests.get(serial)
            if req is not None and not req.finished:
                self._abort_in_batcher(req)
        self.steps += 1
        out = self.batcher.step() if self.batcher.has_work() else {}
        for serial in [s for s, r in self.requests.items() if r.finished]:
            self._by_id.pop(self.requests.pop(serial).request_id, None)
        return out

    def _abort_in_batcher(self, req: SeqRequest):
        b = self.batcher
        if req in b.waiting:
            b.waiting.remove(req)
        elif b._prefilling is not None and b._prefilling[0] is req:
            b._prefilling = None
            b.cache.set_active(req.slot, False)
            b.free_slots.append(req.slot)
        elif b.running.get(req.slot) is req:
            b.cache.set_active(req.slot, False)
            b.running.pop(req.slot, None)
            b.free_slots.append(req.slot)
        req.finished, req.finish_reason = True, "abort"

    def fail_all(self, reason: str = "error"):
        return self.batcher.fail_all(reason)

    @property
    def on_token(self):
        return self.batcher.on_token

    @on_token.setter
    def on_token(self, fn):
        self.batcher.on_token = fn

    def follow(self):
        """Non-source ranks: mirror the source's steps until it says stop."""
        while not self.stopped:
            self.step()

    def shutdown(self):
        """src: release the followers."""
        if self.is_src and not self.stopped:
            self.step(stop=True)
