"""Schedule of the engine pipeline (bigdl_amd/parallel/pipeline.py::EnginePipeline.generate) on timed stand-ins: every
stage is a thread whose "engine" sleeps for a fixed stage time per step and whose link is a set of queues.  What is
checked is the order of waits, which is what decides whether a stage idles: with as many micro-batches as stages the
first stage must spend almost no time waiting for tokens."""
import queue
import threading
import time
import types

import pytest
import torch

from bigdl_amd.parallel.pipeline import FLAG_HALVES, EnginePipeline

HIDDEN, VOCAB, STAGE_S = 8, 32, 0.02


class _Handle:
    def __init__(self, fn=None):
        self.fn = fn

    def wait(self):
        if self.fn is not None:
            self.fn()
            self.fn = None
        return True


class _Link:
    def __init__(self, stage, n, wires):
        self.stage, self.n, self.wires = stage, n, wires
        self.first, self.last = stage == 0, stage == n - 1
        self.head, self.host_staged, self.fwd = 0, True, None

    def _put(self, key, t):
        self.wires[key].put(t.clone())
        return _Handle()

    def _get(self, key, t):
        return _Handle(lambda: t.copy_(self.wires[key].get(timeout=30)))

    def send_hidden(self, t):
        return self._put(("h", self.stage + 1), t)

    def recv_hidden(self, t):
        return self._get(("h", self.stage), t)

    def send_tokens(self, t):
        return self._put(("t", 0), t)

    def recv_tokens(self, t):
        return self._get(("t", 0), t)

    def broadcast_from_head(self, t):
        if self.first:
            for s in range(1, self.n):
                self.wires[("b", s)].put(t.clone())
            return t
        return self.wires[("b", self.stage)].get(timeout=30)


class _Engine:
    """Every step costs STAGE_S of wall time; the "model" adds 1 to a running value so the tokens are predictable."""

    def __init__(self, link):
        self.link = link
        self.cfg = types.SimpleNamespace(hidden_size=HIDDEN)
        self.device = torch.device("cpu")
        self.use_graph = False
        self.cache, self._graphs, self._static = None, {}, {}

    def ensure_cache(self, b, n, fp8=False):
        self.cache = types.SimpleNamespace(reset=lambda: None)

    def _work(self):
        time.sleep(STAGE_S)

    def prefill(self, ids, hidden=None):
        self._work()
        if self.link.first:
            hidden = ids.sum(1).float().view(-1, 1, 1).expand(-1, ids.shape[1], HIDDEN).half()
        hidden = hidden + 1
        if self.link.last:
            return torch.nn.functional.one_hot(hidden[..., 0].long() % VOCAB, VOCAB).float()
        return hidden

    def _get_graph(self, b, q, chain):
        if b not in self._static:
            self._static[b] = {"msg": torch.zeros(b * HIDDEN + FLAG_HALVES, dtype=torch.float16),
                               "tokens": torch.zeros(b, 1, dtype=torch.int64), "next": torch.zeros(b, dtype=torch.int64)}
        return None, self._static[b]

    def decode(self, tokens, q, chain=False):
        self._work()
        for b, st in self._static.items():
            body = st["msg"][:b * HIDDEN].view(b, HIDDEN)
            if self.link.first:
                body.copy_(st["tokens"].half().expand(b, HIDDEN))
            body += 1
            if self.link.last:
                st["next"].copy_(body[:, 0].long() % VOCAB)


def _run(stages, micro_batches, new_tokens=24, eos=None):
    wires = {}
    for s in range(stages + 1):
        wires[("h", s)], wires[("b", s)] = queue.Queue(), queue.Queue()
    wires[("t", 0)] = queue.Queue()
    ids = torch.arange(2 * stages * 3).view(2 * stages, 3) % 5
    pipes, outs, errs = [None] * stages, [None] * stages, []

    def stage(s):
        try:
            link = _Link(s, stages, wires)
            pipes[s] = EnginePipeline(_Engine(link), link)
            # every lane needs its own stand-in state (the real engine gets that from its shallow copies too)
            outs[s] = pipes[s].generate(ids, max_new_tokens=new_tokens, micro_batches=micro_batches, eos_token_id=eos)
        except Exception as e:      # surfaces in the main thread
            errs.append(e)
    th = [threading.Thread(target=stage, args=(s,)) for s in range(stages)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    assert not errs, errs
    return pipes, outs, ids


@pytest.mark.parametrize("stages", [2, 4])
def test_first_stage_does_not_idle_with_one_micro_batch_per_stage(stages):
    pipes, outs, ids = _run(stages, stages)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    # tokens: each pass through the pipeline adds `stages` to the running value
    b = ids.shape[0]
    want = (ids.sum(1) + stages) % VOCAB
    assert torch.equal(outs[0][:, 3], want)
    assert torch.equal(outs[0][:, 4], (want + stages) % VOCAB)
    head = pipes[0]
    idle = head.wait_s / head.decode_wall_s
    assert idle < 0.15, f"first stage waited {idle:.0%} of the decode phase"
    # and the whole decode phase took about steps x lanes x stage time, not the (2n - 1) / n of a per-step barrier
    ideal = 23 * stages * STAGE_S
    assert head.decode_wall_s < 1.25 * ideal, (head.decode_wall_s, ideal)


def test_one_micro_batch_leaves_the_first_stage_waiting():
    """The contrast case (nothing to overlap with): the measurement does see idle time when there is some."""
    pipes, outs, _ = _run(3, 1, new_tokens=12)
    assert pipes[0].wait_s / pipes[0].decode_wall_s > 0.5


def test_stop_flag_reaches_every_stage_in_the_same_step():
    pipes, outs, ids = _run(3, 3, new_tokens=40, eos=list(range(VOCAB)))        # every token is an EOS
    assert all(torch.equal(o, outs[0]) for o in outs)
    assert outs[0].shape[1] == ids.shape[1] + 1                                 # trimmed to the first EOS
    assert len({p.ticks_busy for p in pipes}) == 1 and pipes[0].ticks_busy < 3 * 12
