"""Pipeline stages of the native decode engine.

A stage is a ``NativeDecoder`` that owns a contiguous range of decoder layers (plus the embedding on the first stage
and the final norm + lm_head on the last) and a KV cache for exactly those layers.  Every stage replays ONE hipGraph per
token; between stages a [B, hidden] fp16 message crosses one xGMI link.  Capability: the reference's pipeline-parallel
generation (python/llm/src/ipex_llm/transformers/pipeline_parallel.py:166-235 layer split, :300-445 generate loop with
``dist.send`` / ``dist.recv`` of the hidden states and a broadcast of the next token).

What is different here, and why:

* the transport runs on two communicators: hidden states travel forward on one, the sampled tokens travel from the last
  stage straight back to the first on the other.  RCCL executes the operations of one communicator in issue order, so a
  single communicator would chain "send micro-batch 1 forward" behind "receive the token of micro-batch 0" and stall the
  ring; a broadcast of every token would make the middle stages wait for the tail.  With two communicators the
  forward chain is acyclic (the tail never sends on it) and nothing else synchronises the stages.
* ``micro_batches`` sequences groups are in flight at once (shallow copies of the stage share the weights and own a
  cache + graph each).  The first stage takes a micro-batch's token right before that micro-batch's next step and
  waits for nothing else, so with as many micro-batches as stages every GPU is busy every tick
  (tests/test_pipeline_schedule_cpu.py measures the first stage's idle share on a timed stand-in).
* sends are asynchronous: a message buffer is only waited on right before its next reuse, one full pipeline round
  later.
* the stop decision (all sequences hit EOS) is taken on the first stage and rides in 8 spare halves at the end of the
  hidden-state message; stages look at it every ``STOP_CHECK_EVERY`` steps, so no stage synchronises with its host per
  token.

The module-path equivalents (``pipeline_parallel``, ``PPModelWorker``) live in
bigdl_amd/transformers/pipeline_parallel.py and use the same transport (bigdl_amd/parallel/p2p.py).
"""
from __future__ import annotations

import copy
import dataclasses
import time
from typing import List, Optional

import torch
import torch.distributed as dist

from .p2p import PipelineLink  # noqa: F401
from ..transformers.pipeline_parallel import (PPModelWorker, init_pipeline_parallel, layer_range,  # noqa: F401
                                              pipeline_parallel, pipeline_parallel_generate)

STOP_CHECK_EVERY = 4
FLAG_HALVES = 8          # tail of every hidden-state message: [stop, 7 spare] (fp16)


def make_stage(engine, stage: int, stages: int):
    """Mark ``engine`` (which already holds only its own layers) as stage ``stage`` of ``stages``."""
    engine.stage_first = stage == 0
    engine.stage_last = stage == stages - 1
    if not engine.stage_first and not (engine.stage_last and engine.cfg.tie_word_embeddings):
        engine.embed = None
    if not engine.stage_last:
        engine.lm_head = None
    return engine


def split_engine(engine, stage: int, stages: int):
    """The stage-``stage`` slice of a whole engine (weights are shared, not copied)."""
    from ..models.decoder import NativeDecoder
    start, end = layer_range(len(engine.layers), stages, stage)
    cfg = dataclasses.replace(engine.cfg, num_layers=end - start)
    part = NativeDecoder(cfg, engine.embed, engine.layers[start:end], engine.final_norm, engine.lm_head,
                         device=engine.device, final_norm_b=engine.final_norm_b, pos_embed=engine.pos_embed,
                         embed_norm=engine.embed_norm)
    part.use_graph = engine.use_graph
    return make_stage(part, stage, stages)


def build_random_stage(cfg, qtype, stage: int, stages: int, device="cuda", seed: int = 0):
    """Random-init weights of stage ``stage`` only (a 70B model is never materialised on one GPU)."""
    from ..models.decoder import build_random_decoder
    start, end = layer_range(cfg.num_layers, stages, stage)
    local = dataclasses.replace(cfg, num_layers=end - start)
    return make_stage(build_random_decoder(local, qtype, device=device, seed=seed + stage), stage, stages)


class EnginePipeline:
    """Greedy generation over engine stages.  Every rank of the pipeline constructs this around its own stage and calls
    ``generate`` with the same ``input_ids``; the returned ids are identical on every rank."""

    def __init__(self, stage_engine, link: Optional[PipelineLink] = None):
        self.engine = stage_engine
        self.link = link or PipelineLink()
        self.cfg = stage_engine.cfg
        self.device = stage_engine.device
        self._lanes = [stage_engine]          # one engine view per micro-batch in flight
        self.first_cost = self.rest_cost_mean = self.peak_memory = 0.0
        self.first_token_time = 0.0
        self.last_token_time: List[float] = []
        self.ticks_busy = 0                   # stage steps executed by this rank in the last generate()

    @property
    def use_graph(self):
        return self.engine.use_graph

    @use_graph.setter
    def use_graph(self, v):
        for e in self._lanes:
            e.use_graph = v

    def weight_bytes(self):
        return self.engine.weight_bytes()

    def _lane(self, i):
        while len(self._lanes) <= i:
            e = copy.copy(self.engine)
            e.cache, e._graphs, e._static = None, {}, {}
            self._lanes.append(e)
        return self._lanes[i]

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, max_new_tokens: int = 32, min_new_tokens: int = 0, do_sample=False,
                 eos_token_id=None, pad_token_id=None, kv_fp8=False, micro_batches: int = 1, streamer=None, **unused):
        if do_sample:
            raise NotImplementedError("pipelined generation is greedy (as the reference's pipeline_parallel_generate)")
        link, dev = self.link, self.device
        input_ids = input_ids.to(dev)
        bsz, s = input_ids.shape
        micro_batches = max(1, min(micro_batches, bsz))
        chunks = list(input_ids.chunk(micro_batches))
        nmb = len(chunks)
        lanes = [self._lane(i) for i in range(nmb)]
        hidden = self.cfg.hidden_size
        eos = set([eos_token_id] if isinstance(eos_token_id, int) else (eos_token_id or []))
        for e, ids in zip(lanes, chunks):
            e.ensure_cache(ids.shape[0], s + max_new_tokens + 1, fp8=kv_fp8)
            e.cache.reset()
        sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
        sync()
        if dev.type == "cuda":
            torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        self.ticks_busy = 0
        pending = [None] * nmb                # send in flight per lane (handle, tensor kept alive)
        toks: List[List[torch.Tensor]] = [[] for _ in range(nmb)]   # first stage: tokens per lane, one [b] per step

        # ---- prompt wave: micro-batch after micro-batch through the stages
        for m, (e, ids) in enumerate(zip(lanes, chunks)):
            b = ids.shape[0]
            if link.first:
                out = e.prefill(ids)
            else:
                buf = torch.empty(b, s, hidden, dtype=torch.float16, device=dev)
                link.recv_hidden(buf).wait()
                out = e.prefill(None, hidden=buf)
            self.ticks_busy += 1
            if link.last:
                first_tok = out[:, -1, :].argmax(-1)
                if link.n > 1:
                    pending[m] = (link.send_tokens(first_tok), first_tok)
                else:
                    toks[m].append(first_tok)
            else:
                out = out.contiguous()
                pending[m] = (link.send_hidden(out), out)
        times = []
        if link.first and link.n > 1:
            for m, ids in enumerate(chunks):
                t = torch.empty(ids.shape[0], dtype=torch.int64, device=dev)
                link.recv_tokens(t).wait()
                toks[m].append(t)
        if link.first:
            sync()
            times.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            if streamer is not None:
                streamer.put(input_ids.cpu())
                streamer.put(torch.cat([t[-1] for t in toks]).view(bsz, 1).cpu())
        streamed = 1                          # generated positions already handed to the streamer

        # ---- decode steps: one graph replay per (step, micro-batch) on every stage.  The first stage takes lane m's
        # token right before lane m's next step and nothing else waits: while it waits for lane 0, the later stages are
        # still working on lanes 1.., so with as many lanes as stages no stage idles.  No host synchronisation per
        # step; step times come from events.
        unfinished = [torch.ones(c.shape[0], dtype=torch.bool) for c in chunks]
        scanned = [0] * nmb                   # tokens of each lane already looked at for EOS
        stop_host = False
        steps_done = 1
        on_gpu = dev.type == "cuda"
        marks = []                            # first stage: one event (GPU) / host time (CPU) per step
        self.wait_s = 0.0                     # host time this stage spent blocked on a receive (decode phase)
        loop_t0 = time.perf_counter()
        if on_gpu:
            loop_ev = torch.cuda.Event(enable_timing=True)
            loop_ev.record()

        def collect(m):
            t = torch.empty(chunks[m].shape[0], dtype=torch.int64, device=dev)
            tic = time.perf_counter()
            link.recv_tokens(t).wait()
            self.wait_s += time.perf_counter() - tic
            toks[m].append(t)

        def scan_eos(m):
            for j in range(scanned[m], len(toks[m])):
                if j + 1 >= max(min_new_tokens, 1):
                    for i, t in enumerate(toks[m][j].tolist()):
                        if t in eos:
                            unfinished[m][i] = False
            scanned[m] = len(toks[m])

        for step in range(1, max_new_tokens):
            check = step % STOP_CHECK_EVERY == STOP_CHECK_EVERY - 1
            stop_seen = False
            for m, e in enumerate(lanes):
                b = chunks[m].shape[0]
                g, st = e._get_graph(b, 1, False)
                msg = st["msg"]
                if pending[m] is not None:
                    pending[m][0].wait()
                    pending[m] = None
                if link.first:
                    if link.n > 1 and len(toks[m]) < step:
                        collect(m)
                    if eos and check:            # (host read of the tokens: only every few steps)
                        scan_eos(m)
                        stop_host = m == nmb - 1 and not any(bool(u.any()) for u in unfinished)
                    st["tokens"].copy_(toks[m][-1].view(b, 1))
                    if stop_host:
                        msg[-FLAG_HALVES:].fill_(1.0)
                else:
                    tic = time.perf_counter()
                    link.recv_hidden(msg).wait()
                    self.wait_s += time.perf_counter() - tic
                e.decode(None, 1, chain=False)
                self.ticks_busy += 1
                if link.last:
                    nxt = st["next"].clone()
                    if link.n > 1:
                        pending[m] = (link.send_tokens(nxt), nxt)
                    else:
                        toks[m].append(nxt)
                else:
                    pending[m] = (link.send_hidden(msg), msg)
                if check and float(msg[-FLAG_HALVES]) != 0.0:      # (host read: only every few steps)
                    stop_seen = True
            steps_done += 1
            if link.first:
                if streamer is not None:         # one step behind: the tokens every lane has already delivered
                    have = min(len(t) for t in toks)
                    while streamed < have:
                        streamer.put(torch.cat([t[streamed] for t in toks]).view(bsz, 1).cpu())
                        streamed += 1
                if on_gpu:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    marks.append(ev)
                else:
                    marks.append(time.perf_counter())
            if stop_seen:
                break
        if link.first and link.n > 1:
            for m in range(nmb):
                if len(toks[m]) < steps_done:
                    collect(m)
        for p in pending:
            if p is not None:
                p[0].wait()
        if link.first:
            if streamer is not None:
                while streamed < steps_done:
                    streamer.put(torch.cat([t[streamed] for t in toks]).view(bsz, 1).cpu())
                    streamed += 1
            sync()
            if on_gpu and marks:
                prev = loop_ev
                for mk in marks:
                    times.append(prev.elapsed_time(mk) * 1e-3)
                    prev = mk
            elif marks:
                prev = t0
                for mk in marks:
                    times.append(mk - prev)
                    prev = mk
        self.decode_wall_s = time.perf_counter() - loop_t0
        for e in lanes:                       # a flag left in a message buffer must not stop the next call
            for st in e._static.values():
                st["msg"][-FLAG_HALVES:].zero_()

        # ---- result: assembled on the first stage, shared with the others (one collective per call)
        if link.first:
            gen = torch.cat([torch.stack(t, 1) for t in toks], 0)          # [B, steps]
            if eos:
                host = gen.tolist()
                keep = gen.shape[1]
                for i, row in enumerate(host):
                    hit = [j for j, t in enumerate(row) if t in eos and j + 1 >= max(min_new_tokens, 1)]
                    if hit and pad_token_id is not None:
                        gen[i, hit[0] + 1:] = pad_token_id
                ends = [next((j for j, t in enumerate(row) if t in eos and j + 1 >= max(min_new_tokens, 1)), keep - 1)
                        for row in host]
                gen = gen[:, : max(ends) + 1]
            out_ids = torch.cat([input_ids, gen], 1)
            shape = torch.tensor([out_ids.shape[1]], dtype=torch.int64, device=dev)
        else:
            shape = torch.zeros(1, dtype=torch.int64, device=dev)
        if link.n > 1:
            width = int(link.broadcast_from_head(shape))
            if not link.first:
                out_ids = torch.empty(bsz, width, dtype=torch.int64, device=dev)
            out_ids = link.broadcast_from_head(out_ids)
            stats = torch.tensor([times[0] if times else 0.0, sum(times[1:]) / max(1, len(times) - 1)], dtype=torch.float64,
                                 device=dev)
            first, rest = link.broadcast_from_head(stats).tolist()
        else:
            first, rest = times[0], sum(times[1:]) / max(1, len(times) - 1)
        if streamer is not None and link.first:
            streamer.end()
        self.first_cost = self.first_token_time = first
        self.rest_cost_mean = rest
        self.last_token_time = times[1:]
        if dev.type == "cuda":
            self.peak_memory = torch.cuda.max_memory_reserved() / (1024 ** 3)
        return out_ids
