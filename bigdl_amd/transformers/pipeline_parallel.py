"""Pipeline parallelism: each rank keeps a contiguous range of decoder layers; hidden states travel rank to rank
with point-to-point sends (one xGMI hop between neighbouring GPUs), the last rank produces the token and
broadcasts it.

Capability map to the reference python/llm/src/ipex_llm/transformers/pipeline_parallel.py:
``init_pipeline_parallel`` :108-112, dummy layers :52-105, ``pipeline_parallel`` :166-235, the ``generate`` front door
:238-297, ``pipeline_parallel_generate`` :300-445 (first/next token timing, eos handling), ``PPModelWorker`` :482-930
(batched serving across stages).

Differences by design: layers that live elsewhere are dropped from the module list instead of being replaced by
per-architecture dummy blocks, and the local layers are re-indexed from 0 so every rank's KV cache holds exactly its
own layers; a 288 GB MI355X rarely needs pipeline stages for capacity, so the serving worker keeps several
micro-batches in flight to hide the pipeline bubble rather than to save memory.
"""
from __future__ import annotations

import asyncio
import time
import types
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn

from ..utils.common import invalidInputError


class DummyLayer(nn.Module):
    """Stands in for a module owned by another stage (embedding, final norm, lm_head): returns its input."""

    def __init__(self, *args):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(0), requires_grad=False)

    def forward(self, x, *args, **kwargs):
        return x


class Dummy_DecoderLayer(nn.Module):
    def __init__(self, *args):
        super().__init__()
        self.input_layernorm = DummyLayer()
        self.mlp = DummyLayer()

    def forward(self, hidden_states, *args, **kwargs):
        return hidden_states


Dummy_MLPLayer = DummyLayer
Dummy_GLMBlock = Dummy_DecoderLayer


def init_pipeline_parallel(backend: Optional[str] = None):
    """Join the process group (RCCL on GPUs, gloo otherwise); rendezvous defaults to 127.0.0.1:29500."""
    from ..parallel.comm import init_distributed
    return init_distributed(backend)


def layer_range(num_layers: int, stages: int, rank: int):
    """Contiguous split; the last stage takes the remainder (it also holds the norm + lm_head)."""
    per = num_layers // stages
    start = per * rank
    end = num_layers if rank == stages - 1 else start + per
    return start, end


def _decoder_stack(model):
    for path in ("model", "transformer", "model.language_model"):
        m = model
        ok = True
        for part in path.split("."):
            m = getattr(m, part, None)
            if m is None:
                ok = False
                break
        if ok:
            for name in ("layers", "h", "blocks"):
                if isinstance(getattr(m, name, None), nn.ModuleList):
                    return m, name
    invalidInputError(False, "pipeline_parallel: cannot find the decoder layer list of this model")


def pipeline_parallel(model, pipeline_parallel_stages, torch_dtype=torch.float32, rank=None, device=None):
    """Keep this rank's layer range (plus embedding on the first and norm + lm_head on the last stage) and move it
    to this rank's GPU."""
    rank = dist.get_rank() if rank is None else rank
    stages = pipeline_parallel_stages
    base, lname = _decoder_stack(model)
    layers = getattr(base, lname)
    n = len(layers)
    invalidInputError(n >= stages, "more pipeline stages than decoder layers")
    start, end = layer_range(n, stages, rank)
    local = nn.ModuleList([layers[i] for i in range(start, end)])
    for i, layer in enumerate(local):          # caches are indexed by the attention modules' layer_idx
        for m in layer.modules():
            if hasattr(m, "layer_idx"):
                m.layer_idx = i
    setattr(base, lname, local)
    if hasattr(model.config, "layer_types") and model.config.layer_types is not None:
        model.config.layer_types = list(model.config.layer_types[start:end])
    if rank != 0:
        for name in ("embed_tokens", "wte", "word_embeddings"):
            if hasattr(base, name):
                setattr(base, name, DummyLayer())
    if rank != stages - 1:
        for name in ("norm", "ln_f", "final_layernorm"):
            if hasattr(base, name):
                setattr(base, name, DummyLayer())
        if hasattr(model, "lm_head"):
            model.lm_head = DummyLayer()
    object.__setattr__(model, "pipeline_parallel_stages", stages)
    object.__setattr__(model, "layer_start", start)
    object.__setattr__(model, "layer_end", end)
    object.__setattr__(model, "num_layers", n)
    object.__setattr__(model, "pp_rank", rank)
    if device is None and torch.cuda.is_available():
        device = torch.device("cuda", rank % torch.cuda.device_count())
    if device is not None and torch.device(device).type != "cpu":
        model = model.to(device)
    if not hasattr(model, "_pp_plain_generate"):
        object.__setattr__(model, "_pp_plain_generate", model.generate)
        model.generate = types.MethodType(generate, model)
    model.pipeline_parallel_generate = types.MethodType(pipeline_parallel_generate, model)
    return model


def generate(self, inputs=None, generation_config=None, streamer=None, **kwargs):
    """``model.generate`` of a pipelined model: greedy pipelined decoding."""
    if getattr(self, "pipeline_parallel_stages", 0) >= 1:
        for k in ("do_sample", "num_beams"):
            if kwargs.get(k) not in (None, False, 1):
                invalidInputError(False, f"pipeline parallel generation is greedy; got {k}={kwargs[k]}")
        kwargs.pop("do_sample", None)
        kwargs.pop("num_beams", None)
        max_new_tokens = kwargs.pop("max_new_tokens", None)
        if max_new_tokens is None:
            ml = kwargs.pop("max_length", None) or getattr(generation_config or self.generation_config, "max_length", 20)
            ids = inputs if inputs is not None else kwargs.get("input_ids")
            max_new_tokens = max(1, ml - ids.shape[1])
        return self.pipeline_parallel_generate(inputs=inputs, max_new_tokens=max_new_tokens,
                                               generation_config=generation_config, streamer=streamer, **kwargs)
    return self._pp_plain_generate(inputs=inputs, generation_config=generation_config, streamer=streamer, **kwargs)


def _stage_forward(model, input_ids, inputs_embeds, cache, position_ids, attention_mask=None):
    """Run this stage: tokens -> hidden on the first stage, hidden -> hidden in the middle, hidden -> logits on
    the last.  Returns (tensor, cache)."""
    kw = dict(past_key_values=cache, use_cache=True, position_ids=position_ids)
    if attention_mask is not None:
        kw["attention_mask"] = attention_mask
    if input_ids is not None:
        out = model(input_ids=input_ids, **kw)
    else:
        out = model(inputs_embeds=inputs_embeds, **kw)
    return out.logits, out.past_key_values


@torch.no_grad()
def pipeline_parallel_generate(self, inputs: Optional[torch.Tensor] = None, max_new_tokens: int = 32,
                               generation_config=None, streamer=None, **kwargs):
    gc = generation_config or self.generation_config
    input_ids = inputs if inputs is not None else kwargs.pop("input_ids")
    attention_mask = kwargs.pop("attention_mask", None)
    stages = self.pipeline_parallel_stages
    rank = dist.get_rank()
    pre_rank, next_rank = (rank - 1) % stages, (rank + 1) % stages
    dev = next(p for p in self.parameters() if p.numel() > 0).device
    dtype = next(p for p in self.parameters() if p.numel() > 0 and p.dtype.is_floating_point and p.dim() == 1).dtype
    input_ids = input_ids.to(dev)
    bs = input_ids.shape[0]
    eos = kwargs.pop("eos_token_id", getattr(gc, "eos_token_id", None))
    eos = [eos] if isinstance(eos, int) else eos
    pad = kwargs.pop("pad_token_id", getattr(gc, "pad_token_id", None))
    if eos is not None and pad is None:
        pad = eos[0]
    eos_t = torch.tensor(eos, device=dev) if eos else None
    # lm_head keeps only the last position for long inputs: every stage but the last must pass full hidden states
    lm_head = getattr(self, "lm_head", None)
    self.first_token_time = 0
    self.next_token_time = []
    output_ids = input_ids.clone()
    unfinished = torch.ones(bs, dtype=torch.long, device=dev)
    cache = None
    cur = input_ids
    past = 0
    if streamer is not None:
        streamer.put(input_ids.cpu())
    for step in range(max_new_tokens):
        tic = time.time()
        n = cur.shape[1]
        pos = torch.arange(past, past + n, device=dev)[None].expand(bs, -1)
        am = None
        if attention_mask is not None:
            am = torch.cat([attention_mask.to(dev), torch.ones(bs, past + n - attention_mask.shape[1], dtype=attention_mask.dtype,
                                                               device=dev)], 1) if past + n > attention_mask.shape[1] else attention_mask.to(dev)
        if rank == 0:
            out, cache = _stage_forward(self, cur, None, cache, pos, am)
        else:
            hidden = torch.empty(bs, n, self.config.hidden_size, device=dev, dtype=dtype)
            dist.recv(hidden, src=pre_rank)
            out, cache = _stage_forward(self, None, hidden, cache, pos, am)
        if rank == stages - 1:
            next_ids = torch.argmax(out[:, -1:, :], dim=-1)
            dist.broadcast(next_ids, src=rank)
        else:
            dist.send(out.to(dtype).contiguous(), dst=next_rank)
            next_ids = torch.empty((bs, 1), device=dev, dtype=torch.int64)
            dist.broadcast(next_ids, src=stages - 1)
        past += n
        tok = next_ids.squeeze(1)
        if eos_t is not None:
            tok = tok * unfinished + pad * (1 - unfinished)
        cur = tok.view(bs, 1)
        output_ids = torch.cat([output_ids, cur], dim=-1)
        if streamer is not None:
            streamer.put(cur.cpu())
        if dev.type == "cuda":
            torch.cuda.synchronize()
        toc = time.time()
        if step == 0:
            self.first_token_time = toc - tic
        else:
            self.next_token_time.append(toc - tic)
        if eos_t is not None:
            unfinished = unfinished * (tok.unsqueeze(0) != eos_t.unsqueeze(1)).all(0).long()
            if int(unfinished.max()) == 0:
                break
    if streamer is not None:
        streamer.end()
    self.first_cost = self.first_token_time
    self.rest_cost_mean = sum(self.next_token_time) / max(1, len(self.next_token_time))
    return output_ids


# ---------------------------------------------------------------------------------------------
# serving across stages
# ---------------------------------------------------------------------------------------------
class PPConfig:
    """Stage coordinates of a serving worker."""

    def __init__(self, pp_rank: int, pp_world_size: int) -> None:
        self.pp_rank = pp_rank
        self.pp_world_size = pp_world_size
        self.is_head = pp_rank == 0
        self.is_tail = pp_rank == pp_world_size - 1


class BatchTask:
    """One micro-batch travelling round the pipeline."""

    def __init__(self, batch_id: str, request_ids: List[str], max_tokens: int, batch_size: int, input_len: int,
                 prompt_lengths: List[int], stopped: bool = False, prefilled_index: int = 0, partial_prefilling: int = 0):
        self.batch_id, self.request_ids, self.max_tokens, self.batch_size = batch_id, request_ids, max_tokens, batch_size
        self.input_len, self.prompt_lengths, self.stopped = input_len, prompt_lengths, stopped
        self.prefilled_index, self.partial_prefilling = prefilled_index, partial_prefilling

    def dict(self):
        return dict(self.__dict__)


def make_attention_mask(prompt_lengths, device):
    max_len = max(prompt_lengths)
    mask = torch.zeros(len(prompt_lengths), max_len, dtype=torch.int64, device=device)
    for i, n in enumerate(prompt_lengths):
        mask[i, max_len - n:] = 1          # left padding
    return mask


KIND_STEP, KIND_FREE, KIND_STOP = 1, 2, 3
HEADER_FIXED = 8          # [kind, serial, batch_size, n_tokens, step_no, max_tokens, piece start, piece count] + prompt lengths


class PPModelWorker:
    """Continuous-batching worker for one pipeline stage.

    The head stage admits requests (up to ``max_num_seqs`` sequences per micro-batch; ``max_prefilled_seqs`` > 0 caps the
    sequences a newly admitted micro-batch may hold, which bounds the size of one prompt pass), runs its layers and sends
    the hidden states on; the tail stage picks the tokens and sends them straight back to the head, which appends them
    and schedules the next step.  Up to ``pp_world_size`` micro-batches are in flight, so in steady state every stage has
    a micro-batch to work on at every tick.

    Control travels with the data: each hidden-state message is preceded by a small integer header (kind, serial
    number, shapes, step), so no stage takes part in a collective or an object broadcast per token; a FREE header drops
    a finished micro-batch's cache on every stage and a STOP header ends the serving loops.  Sends are asynchronous and
    their buffers are only released once the transfer has completed.
    """

    def __init__(self, checkpoint, rank, world_size, low_bit, max_num_seqs, max_prefilled_seqs=0,
                 torch_dtype=torch.float16, model=None, tokenizer=None, link=None):
        self.dtype = torch_dtype
        self.model = model if model is not None else self.load_model(checkpoint, world_size, low_bit)
        self.device = next(p for p in self.model.parameters() if p.numel() > 0).device
        self._init_state(rank, world_size, max_num_seqs, max_prefilled_seqs, tokenizer, link)

    def _init_state(self, rank, world_size, max_num_seqs, max_prefilled_seqs, tokenizer, link):
        from collections import deque
        from ..parallel.p2p import PipelineLink
        self.pp_config = PPConfig(rank, world_size)
        self.rank, self.world_size = rank, world_size
        self.pre_rank, self.next_rank = (rank - 1) % world_size, (rank + 1) % world_size
        self.max_num_seqs = max_num_seqs
        self.max_prefilled_seqs = max_prefilled_seqs
        self.tokenizer = tokenizer
        self.link = link if link is not None else (PipelineLink(stage=rank) if world_size > 1 else None)
        self.waiting_requests = asyncio.Queue()
        self.kv_caches = {}                 # batch_id -> cache
        self.positions = {}                 # batch_id -> tokens already in the cache
        self.tokens = {}                    # batch_id -> list of [B, 1] token tensors (head)
        self.token_times = {}
        self.streamer = {}                  # request_id -> asyncio.Queue of text deltas
        self.token_cache = {}
        self.print_len = {}
        self.is_finish = {}
        self.inflight = deque()             # head: micro-batches whose tokens are still to come back, oldest first
        self.on_going_batches = self.inflight
        self.results = {}                   # head: batch_id -> generated ids [B, T] of finished micro-batches
        self.input_ids_dict = {}
        self.attention_mask_dict = {}
        self._sends = deque()               # (work, tensor) of transfers not known to have completed
        self._local_tokens = deque()        # single-stage pipeline: the head is its own tail
        self._serial = 0
        self.stopped = False
        self.trace = []                     # (serial, step_no) of every stage step this rank ran
        self.stats = dict(requests=0, prompt_tokens=0, generated_tokens=0, busy_s=0.0, first_token_s=[], per_token_s=[])
        self.max_inflight = 0
        self.clean_cache()

    def load_model(self, model_path, world_size, low_bit="sym_int4"):
        from .model import AutoModelForCausalLM
        try:
            model = AutoModelForCausalLM.load_low_bit(model_path, pipeline_parallel_stages=world_size,
                                                      torch_dtype=self.dtype)
        except Exception:
            model = AutoModelForCausalLM.from_pretrained(model_path, load_in_low_bit=low_bit, torch_dtype=self.dtype,
                                                         optimize_model=True, pipeline_parallel_stages=world_size)
        return model.eval()

    def clean_cache(self):
        self.kv_caches.clear()
        self.positions.clear()

    def clear_batch(self, cur_id):
        for d in (self.kv_caches, self.positions, self.tokens, self.token_times, self.input_ids_dict,
                  self.attention_mask_dict):
            d.pop(cur_id, None)

    @torch.no_grad()
    def model_step(self, inp, cur_batch: BatchTask):
        """One stage step for one micro-batch.  ``inp``: token ids [B, n] on the head, hidden states otherwise.
        Returns next-token ids [B, 1] on the tail, hidden states elsewhere."""
        bid = cur_batch.batch_id
        cache = self.kv_caches.get(bid)
        past = self.positions.get(bid, 0)
        n = inp.shape[1]
        bs = inp.shape[0]
        am = self.attention_mask_dict.get(bid)
        if am is None and past == 0 and len(set(cur_batch.prompt_lengths)) > 1:
            am = make_attention_mask(cur_batch.prompt_lengths, self.device)
        if am is not None:
            grow = past + n - am.shape[1]
            if grow > 0:
                am = torch.cat([am, torch.ones(bs, grow, dtype=am.dtype, device=am.device)], 1)
            self.attention_mask_dict[bid] = am
            pos = (am.cumsum(-1) - 1).clamp_min(0)[:, past:past + n]
        else:
            pos = torch.arange(past, past + n, device=self.device)[None].expand(bs, -1)
        if self.pp_config.is_head:
            out, cache = _stage_forward(self.model, inp, None, cache, pos, am)
        else:
            out, cache = _stage_forward(self.model, None, inp, cache, pos, am)
        self.kv_caches[bid] = cache
        self.positions[bid] = past + n
        if self.pp_config.is_tail:
            return torch.argmax(out[:, -1:, :], dim=-1)
        return out

    def is_initialized(self):
        return True

    # -- transport ------------------------------------------------------------------------------
    def _header_device(self):
        return self.device if self.link is not None and not self.link.host_staged else torch.device("cpu")

    def _post(self, work, tensor):
        """Remember an asynchronous send; drop the ones that have completed (their buffers may be reused now)."""
        self._sends.append((work, tensor))
        while self._sends and self._sends[0][0].is_completed():
            self._sends.popleft()

    def _send_header(self, kind, task: Optional[BatchTask] = None, n_tokens=0, step_no=0, piece=(0, 0)):
        """``piece`` = (first sequence, count) of a partial prompt pass; the lengths that follow are that piece's."""
        h = torch.zeros(HEADER_FIXED + self.max_num_seqs, dtype=torch.int64)
        h[0] = kind
        if task is not None:
            h[1], h[2], h[3], h[4], h[5] = task.serial, task.batch_size, n_tokens, step_no, task.max_tokens
            h[6], h[7] = piece
            lens = task.prompt_lengths[piece[0]: piece[0] + piece[1]] if piece[1] else task.prompt_lengths
            h[HEADER_FIXED: HEADER_FIXED + len(lens)] = torch.tensor(lens)
        h = h.to(self._header_device())
        self._post(self.link.send_hidden(h), h)

    def _recv_header(self):
        h = torch.zeros(HEADER_FIXED + self.max_num_seqs, dtype=torch.int64, device=self._header_device())
        self.link.recv_hidden(h).wait()
        return h.tolist()

    def drain_sends(self):
        while self._sends:
            self._sends.popleft()[0].wait()

    # -- head-side request handling -------------------------------------------------------------
    async def add_request(self, request, prompt_ids: Optional[List[int]] = None, max_tokens: Optional[int] = None):
        """``add_request(request_id, prompt_ids, max_tokens)`` or ``add_request(req)`` with the serving front end's
        request object (serving/fastapi/model_worker.py ``Request``).  Returns the queue of (text delta, finished)."""
        if prompt_ids is None:
            req = request
            request = req.request_id
            prompt_ids = req.input_ids if req.input_ids is not None else self.tokenizer(req.prompt).input_ids
            max_tokens = req.max_new_tokens
        q = self.streamer[request] = asyncio.Queue()
        self.stats["requests"] += 1
        self.stats["prompt_tokens"] += len(prompt_ids)
        await self.waiting_requests.put((request, list(prompt_ids), max_tokens))
        return q

    def finish(self, request_id: str):
        self.streamer.pop(request_id, None)
        self.is_finish.pop(request_id, None)

    def _admit(self):
        limit = self.max_num_seqs
        if self.max_prefilled_seqs and self.max_prefilled_seqs > 0:
            limit = min(limit, self.max_prefilled_seqs)
        reqs = []
        while len(reqs) < limit and not self.waiting_requests.empty():
            reqs.append(self.waiting_requests.get_nowait())
        if not reqs:
            return None, None
        lens = [len(r[1]) for r in reqs]
        mx = max(lens)
        pad = getattr(self.tokenizer, "pad_token_id", None) or 0
        ids = torch.full((len(reqs), mx), pad, dtype=torch.long, device=self.device)
        for i, (_, p, _) in enumerate(reqs):
            ids[i, mx - len(p):] = torch.tensor(p, device=self.device)
        self._serial += 1
        task = BatchTask(batch_id=f"batch_{self._serial}", request_ids=[r[0] for r in reqs],
                         max_tokens=max(r[2] for r in reqs), batch_size=len(reqs), input_len=mx, prompt_lengths=lens)
        task.serial = self._serial
        task.step_no = 0
        return task, ids

    def get_printable_text(self, cur_text, request_id):
        plen = self.print_len.get(request_id, 0)
        if cur_text.endswith("\n"):
            out = cur_text[plen:]
            self.token_cache[request_id] = []
            self.print_len[request_id] = 0
        elif len(cur_text) > 0 and _is_chinese_char(ord(cur_text[-1])):
            out = cur_text[plen:]
            self.print_len[request_id] = plen + len(out)
        else:
            out = cur_text[plen: cur_text.rfind(" ") + 1]
            self.print_len[request_id] = plen + len(out)
        return out

    def _emit(self, task: BatchTask, next_ids, step_no, first=0):
        """Head: record the sampled tokens (``next_ids[j]`` belongs to sequence ``first + j`` of the micro-batch), stream
        text deltas, report whether every sequence they belong to is done."""
        eos = getattr(self.tokenizer, "eos_token_id", None)
        all_done = True
        for i, rid in enumerate(task.request_ids[first: first + len(next_ids)]):
            if self.is_finish.get(rid):
                continue
            t = int(next_ids[i])
            cache = self.token_cache.setdefault(rid, [])
            cache.append(t)
            budget = getattr(task, "request_max", None)
            finished = (eos is not None and t == eos) or step_no + 1 >= (budget[first + i] if budget else task.max_tokens)
            if self.tokenizer is not None:
                text = self.tokenizer.decode(cache, skip_special_tokens=True)
                delta = self.get_printable_text(text, rid) if not finished else text[self.print_len.get(rid, 0):]
            else:
                delta = t
            if rid in self.streamer:          # (a client that went away has had its queue dropped)
                self.streamer[rid].put_nowait((delta, finished))
            self.stats["generated_tokens"] += 1
            if finished:
                self.is_finish[rid] = True
                self.token_cache.pop(rid, None)
                self.print_len.pop(rid, None)
            else:
                all_done = False
        return all_done

    @torch.no_grad()
    def _head_step(self, task: BatchTask, inp: torch.Tensor):
        """Head: run this stage on ``inp`` (prompt ids or the last tokens) and hand the result to the next stage."""
        self.trace.append((task.serial, task.step_no))
        out = self.model_step(inp, task)
        if self.link is None:
            self._local_tokens.append(out)
        else:
            self._send_header(KIND_STEP, task, n_tokens=inp.shape[1], step_no=task.step_no)
            out = out.to(self.dtype).contiguous()
            self._post(self.link.send_hidden(out), out)
        self.inflight.append(task)
        self.max_inflight = max(self.max_inflight, len(self.inflight))

    def _head_collect(self):
        """Head: take the tokens of the oldest micro-batch in flight, stream them, and either step it again or retire
        it.  Returns the finished micro-batch's ids [B, T], or None while it goes on."""
        task = self.inflight.popleft()
        if self.link is None:
            next_ids = self._local_tokens.popleft()
        else:
            next_ids = torch.empty(task.batch_size, 1, dtype=torch.int64, device=self.device)
            self.link.recv_tokens(next_ids).wait()
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

    @torch.no_grad()
    def _serve_one(self):
        """A later stage: take the next message from the previous stage and act on it."""
        h = self._recv_header()
        kind, serial, bs, n, step_no, max_tokens = h[:6]
        task = BatchTask(batch_id=f"batch_{serial}", request_ids=[], max_tokens=max_tokens, batch_size=bs, input_len=n,
                         prompt_lengths=h[HEADER_FIXED: HEADER_FIXED + bs])
        task.serial, task.step_no = serial, step_no
        if kind != KIND_STEP:
            if kind == KIND_FREE:
                self.clear_batch(task.batch_id)
            else:
                self.stopped = True
            if not self.pp_config.is_tail:
                self._send_header(kind, task)
            return kind
        hidden = torch.empty(bs, n, self.model.config.hidden_size, device=self.device, dtype=self.dtype)
        self.link.recv_hidden(hidden).wait()
        self.trace.append((serial, step_no))
        out = self.model_step(hidden, task)
        if self.pp_config.is_tail:
            out = out.contiguous()
            self._post(self.link.send_tokens(out), out)
        else:
            self._send_header(KIND_STEP, task, n_tokens=n, step_no=step_no)
            out = out.to(self.dtype).contiguous()
            self._post(self.link.send_hidden(out), out)
        return kind

    async def process_step(self):
        """One scheduling tick (called repeatedly by the serving loop of every rank).  Head: fill the pipeline with
        newly admitted micro-batches while there is room, then advance the oldest one in flight; returns the ids of a
        micro-batch that finished in this tick, else None.  Other stages: serve one incoming message."""
        if not self.pp_config.is_head:
            return None if self.stopped else self._serve_one()
        if not self.busy():
            await asyncio.sleep(0.002)        # idle: leave the event loop to the HTTP handlers
            return None
        await asyncio.sleep(0)
        tic = time.time()
        try:
            return self._head_tick()
        finally:
            self.stats["busy_s"] += time.time() - tic

    def _head_tick(self):
        while len(self.inflight) < self.world_size and not self.waiting_requests.empty():
            task, ids = self._admit()
            self._head_step(task, ids)
        if not self.inflight:
            return None
        return self._head_collect()

    def busy(self):
        return bool(self.inflight) or not self.waiting_requests.empty()

    def shutdown(self):
        """Head: tell the other stages to leave their serving loops.  Every rank: wait for its sends."""
        if self.pp_config.is_head and self.link is not None and not self.stopped:
            self._send_header(KIND_STOP)
            self.stopped = True
        self.drain_sends()

    async def run_until_idle(self):
        """Head: drive every admitted and waiting request to completion; other stages: serve until STOP."""
        if self.pp_config.is_head:
            while self.busy():
                await self.process_step()
            return dict(self.results)
        while not self.stopped:
            await self.process_step()
        self.drain_sends()
        return {}


def _is_chinese_char(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF
            or 0x2F800 <= cp <= 0x2FA1F)
