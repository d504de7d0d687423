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
from ..transformers.pipeline_parallel import (HEADER_FIXED, KIND_FREE, KIND_STEP, BatchTask, PPModelWorker,  # noqa: F401
                                              init_pipeline_parallel, layer_range, pipeline_parallel,
                                              pipeline_parallel_generate)

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


class EnginePPWorker(PPModelWorker):
    """The continuous-batching pipeline worker (``PPModelWorker``: admission, headers that travel with the data, up to
    ``world_size`` micro-batches in flight, text streaming) with every stage step run by a native engine stage instead
    of HuggingFace modules: a decode step of a micro-batch is one hipGraph replay over that micro-batch's cache slots,
    and a prompt pass runs the admitted prompts packed, without padding.

    Partial prefill: with ``max_prefilled_seqs = P`` a micro-batch of B > P admitted sequences goes through its prompt
    phase in pieces of P sequences.  Every piece travels round the pipeline as a step of its own (other micro-batches'
    decode steps run in between, so one large admission never blocks the stages for a whole B-prompt pass), fills its
    own cache slots on every stage -- the "joined" cache is simply the micro-batch's slot-mode cache once all pieces are
    in -- and returns the first token of its sequences to the head; when the last piece is back the micro-batch starts
    decoding all B sequences together.  Reference: ``max_prefilled_seqs`` / ``partial_prefilling`` / ``cat_kv_cache``
    in python/llm/src/ipex_llm/transformers/pipeline_parallel.py:545-591, 621-700, 807-927.
    """

    def __init__(self, stage_engine, rank, world_size, max_num_seqs, max_prefilled_seqs=0, tokenizer=None, link=None,
                 max_seq_len: int = 2048, kv_fp8=False):
        self.engine = stage_engine
        self.model = None
        self.dtype = torch.float16
        self.device = stage_engine.device
        self.max_seq_len, self.kv_fp8 = max_seq_len, kv_fp8
        self.lanes = {}                     # serial -> [engine view, decoding?]
        self._idle_lanes = []
        self.prompt_passes = 0              # prompt pieces this stage has run
        self._init_state(rank, world_size, max_num_seqs, max_prefilled_seqs, tokenizer, link)

    # -- per-micro-batch engine views -----------------------------------------------------------
    def _lane(self, serial):
        ent = self.lanes.get(serial)
        if ent is None:
            if self._idle_lanes:
                e = self._idle_lanes.pop()
            else:
                e = copy.copy(self.engine)
                e.cache, e._graphs, e._static = None, {}, {}
            e.ensure_cache(self.max_num_seqs, self.max_seq_len, fp8=self.kv_fp8)
            e.cache.enable_slots()
            ent = self.lanes[serial] = [e, False]
        return ent

    def clear_batch(self, cur_id):
        super().clear_batch(cur_id)
        ent = self.lanes.pop(int(str(cur_id).rsplit("_", 1)[-1]), None)
        if ent is not None:
            self._idle_lanes.append(ent[0])

    def clean_cache(self):
        super().clean_cache()
        for ent in getattr(self, "lanes", {}).values():
            self._idle_lanes.append(ent[0])
        if hasattr(self, "lanes"):
            self.lanes.clear()

    # -- stage steps ----------------------------------------------------------------------------
    def _prompt_piece(self, serial, start, lens, prompts=None, hidden=None):
        e = self._lane(serial)[0]
        slots = list(range(start, start + len(lens)))
        self.prompt_passes += 1
        if self.pp_config.is_head:
            out = e.prefill_packed(prompts, slots)
        else:
            out = e.prefill_packed(None, slots, hidden=hidden, lens=lens)
        if self.pp_config.is_tail:
            return out.argmax(-1)                                   # [count] first tokens
        return out                                                  # [rows, hidden]

    def _decode(self, serial, bs, tokens=None):
        ent = self._lane(serial)
        e = ent[0]
        if not ent[1]:                                              # first decode step: every sequence is in its slot
            for sl in range(bs):
                e.cache.set_active(sl, True)
            ent[1] = True
        g, st = e._get_graph(self.max_num_seqs, 1, False)
        if tokens is not None:
            st["tokens"][:bs].copy_(tokens.view(bs, 1))
        return e, st

    # -- head -----------------------------------------------------------------------------------
    def _admit(self):
        reqs = []
        while len(reqs) < self.max_num_seqs and not self.waiting_requests.empty():
            reqs.append(self.waiting_requests.get_nowait())
        if not reqs:
            return None, None
        lens = [len(r[1]) for r in reqs]
        self._serial += 1
        room = [max(1, self.max_seq_len - n - 1) for n in lens]
        task = BatchTask(batch_id=f"batch_{self._serial}", request_ids=[r[0] for r in reqs],
                         max_tokens=max(min(r[2], rm) for r, rm in zip(reqs, room)), batch_size=len(reqs),
                         input_len=max(lens), prompt_lengths=lens)
        task.request_max = [min(r[2], rm) for r, rm in zip(reqs, room)]
        task.serial, task.step_no = self._serial, 0
        task.prompts = [list(r[1]) for r in reqs]
        p = self.max_prefilled_seqs if self.max_prefilled_seqs and self.max_prefilled_seqs > 0 else len(reqs)
        task.pieces = [(i, min(p, len(reqs) - i)) for i in range(0, len(reqs), p)]
        task.next_piece = 0
        task.first_tokens = []
        return task, None

    @torch.no_grad()
    def _head_step(self, task, inp):
        self.trace.append((task.serial, task.step_no))
        if task.next_piece < len(task.pieces):
            start, count = task.pieces[task.next_piece]
            lens = task.prompt_lengths[start:start + count]
            out = self._prompt_piece(task.serial, start, lens, prompts=task.prompts[start:start + count])
            if self.link is None:
                self._local_tokens.append(out.view(-1, 1))
            else:
                self._send_header(KIND_STEP, task, n_tokens=sum(lens), step_no=0, piece=(start, count))
                out = out.contiguous()
                self._post(self.link.send_hidden(out), out)
        else:
            e, st = self._decode(task.serial, task.batch_size, inp)
            e.decode(None, 1, chain=False)
            if self.link is None:
                self._local_tokens.append(st["next"][:task.batch_size].clone().view(-1, 1))
            else:
                self._send_header(KIND_STEP, task, n_tokens=0, step_no=task.step_no)
                # (the lane's message buffer is next written by this micro-batch's next step, a full round trip away)
                self._post(self.link.send_hidden(st["msg"]), st["msg"])
        self.inflight.append(task)
        self.max_inflight = max(self.max_inflight, len(self.inflight))

    def _head_collect(self):
        task = self.inflight.popleft()
        prompt_phase = task.next_piece < len(task.pieces)
        n = task.pieces[task.next_piece][1] if prompt_phase else task.batch_size
        if self.link is None:
            next_ids = self._local_tokens.popleft()
        else:
            next_ids = torch.empty(n, 1, dtype=torch.int64, device=self.device)
            self.link.recv_tokens(next_ids).wait()
        if prompt_phase:
            start = task.pieces[task.next_piece][0]
            task.first_tokens.append(next_ids)
            self._emit(task, next_ids.view(-1).tolist(), 0, first=start)       # first tokens stream as their piece returns
            task.next_piece += 1
            if task.next_piece < len(task.pieces):
                self._head_step(task, None)
                return None
            next_ids = torch.cat(task.first_tokens)
            self.tokens.setdefault(task.batch_id, []).append(next_ids)
            done = all(self.is_finish.get(rid) for rid in task.request_ids)
        else:
            self.tokens.setdefault(task.batch_id, []).append(next_ids)
            done = self._emit(task, next_ids.view(-1).tolist(), task.step_no)
        if done:
            gen = torch.cat(self.tokens[task.batch_id], 1)
            self.results[task.batch_id] = gen
            self.clear_batch(task.batch_id)
            if self.link is not None:
                self._send_header(KIND_FREE, task)
            return gen
        task.step_no += 1
        self._head_step(task, next_ids)
        return None

    # -- later stages ---------------------------------------------------------------------------
    @torch.no_grad()
    def _serve_one(self):
        h = self._recv_header()
        kind, serial, bs, n, step_no, max_tokens, start, count = h[:HEADER_FIXED]
        task = BatchTask(batch_id=f"batch_{serial}", request_ids=[], max_tokens=max_tokens, batch_size=bs, input_len=n,
                         prompt_lengths=h[HEADER_FIXED: HEADER_FIXED + (count or bs)])
        task.serial, task.step_no = serial, step_no
        if kind != KIND_STEP:
            if kind == KIND_FREE:
                self.clear_batch(task.batch_id)
            else:
                self.stopped = True
            if not self.pp_config.is_tail:
                self._send_header(kind, task)
            return kind
        self.trace.append((serial, step_no))
        hidden_size = self.engine.cfg.hidden_size
        if n > 0:                                                   # a prompt piece
            hidden = torch.empty(n, hidden_size, device=self.device, dtype=torch.float16)
            self.link.recv_hidden(hidden).wait()
            lens = task.prompt_lengths[:count]
            out = self._prompt_piece(serial, start, lens, hidden=hidden)
            if self.pp_config.is_tail:
                out = out.view(-1, 1).contiguous()
                self._post(self.link.send_tokens(out), out)
            else:
                task.prompt_lengths = [0] * start + list(lens)      # (_send_header slices the piece out again)
                self._send_header(KIND_STEP, task, n_tokens=n, step_no=0, piece=(start, count))
                out = out.contiguous()
                self._post(self.link.send_hidden(out), out)
            return kind
        e, st = self._decode(serial, bs)
        self.link.recv_hidden(st["msg"]).wait()
        e.decode(None, 1, chain=False)
        if self.pp_config.is_tail:
            out = st["next"][:bs].clone().view(-1, 1)
            self._post(self.link.send_tokens(out), out)
        else:
            self._send_header(KIND_STEP, task, n_tokens=0, step_no=step_no)
            self._post(self.link.send_hidden(st["msg"]), st["msg"])
        return kind
