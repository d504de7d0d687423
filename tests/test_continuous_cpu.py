"""Continuous-batching scheduler logic on a fake engine (slot admission, eviction, per-slot lengths)."""
import torch

from bigdl_amd.serving.continuous import ContinuousBatcher, SeqRequest, generate_batch


class _FakeCache:
    def __init__(self, batch, capacity):
        self.batch, self.capacity = batch, capacity
        self.slot_len = None
        self.active_host = [1] * batch

    def enable_slots(self):
        self.slot_len = [0] * self.batch
        self.active_host = [0] * self.batch

    def set_active(self, slot, on):
        self.active_host[slot] = 1 if on else 0
        if not on:
            self.slot_len[slot] = 0


class _FakeEngine:
    """Next token = (sum of the slot's tokens so far) % vocab: depends on the whole per-slot history, so any mix-up
    between slots or a stale cache row changes the output."""
    vocab = 97
    device = torch.device("cpu")

    def ensure_cache(self, batch, needed, fp8=False):
        self.cache = _FakeCache(batch, needed)
        self.hist = [[] for _ in range(batch)]
        self.tokens = torch.zeros(batch, 1, dtype=torch.int64)
        return self.cache

    def _get_graph(self, batch, q_len, chain):
        return None, {"tokens": self.tokens}

    def _logits(self, hist):
        lg = torch.zeros(self.vocab)
        lg[sum(hist) % self.vocab] = 1.0
        return lg

    mixed = 0
    chunks = []

    def prefill(self, ids, last_only=True, slot=None, with_decode=False):
        assert not self.cache.active_host[slot]
        if self.cache.slot_len[slot] == 0:
            self.hist[slot] = []
        assert len(self.hist[slot]) == self.cache.slot_len[slot]      # a later chunk continues where the last one ended
        self.hist[slot] += ids[0].tolist()
        self.chunks.append(ids.shape[1])
        self.cache.slot_len[slot] += ids.shape[1]
        first = self._logits(self.hist[slot])[None, None]
        if with_decode:                               # the running slots advance in the same pass
            self.mixed += 1
            return (first,) + self.decode(None, 1, chain=True)
        return first

    packed = 0

    def prefill_packed(self, prompts, slots):
        self.packed += len(prompts)
        return torch.stack([self.prefill(torch.tensor([p]), slot=s)[0, 0] for p, s in zip(prompts, slots)])

    def decode(self, tokens, q_len=1, chain=False):
        lg = torch.zeros(self.cache.batch, self.vocab)
        for s in range(self.cache.batch):
            if self.cache.active_host[s]:
                self.hist[s].append(int(self.tokens[s, 0]))
                self.cache.slot_len[s] += 1
                lg[s] = self._logits(self.hist[s])
        nxt = lg.argmax(-1)
        self.tokens[:, 0] = nxt
        return lg, nxt


def _expected(prompt, n, vocab=97):
    h, out = list(prompt), []
    for _ in range(n):
        t = sum(h) % vocab
        out.append(t)
        h.append(t)
    return out


def test_more_requests_than_slots_all_finish_with_their_own_history():
    prompts = [[i + 1, 2 * i + 3, 5] + [7] * i for i in range(7)]
    outs = generate_batch(_FakeEngine(), prompts, max_new_tokens=6, max_num_seqs=3)
    for p, o in zip(prompts, outs):
        assert o == _expected(p, 6)


def test_piggyback_and_separate_passes_give_the_same_tokens():
    prompts = [[i + 1, 2 * i + 3, 5] + [7] * i for i in range(9)]
    outs = {}
    for on in (True, False):
        eng = _FakeEngine()
        cb = ContinuousBatcher(eng, max_num_seqs=3, max_seq_len=64, piggyback=on)
        reqs = [cb.submit(SeqRequest(str(i), list(p), max_new_tokens=3 + i % 4)) for i, p in enumerate(prompts)]
        cb.run_until_done()
        outs[on] = [r.output_ids for r in reqs]
        assert (eng.mixed > 0) == on and cb.mixed_steps == eng.mixed
    assert outs[True] == outs[False] == [_expected(p, 3 + i % 4) for i, p in enumerate(prompts)]


def test_waiting_prompts_are_packed_into_one_pass():
    prompts = [[i + 1, 2 * i + 3, 5] + [7] * i for i in range(9)]
    eng = _FakeEngine()
    cb = ContinuousBatcher(eng, max_num_seqs=4, max_seq_len=64, max_prefill_per_step=4, max_prefill_tokens=16)
    reqs = [cb.submit(SeqRequest(str(i), list(p), max_new_tokens=4)) for i, p in enumerate(prompts)]
    cb.step()                                         # nothing runs yet: one prompt pass + a packed pass for the rest
    assert len(cb.running) == 4 and eng.packed == 3 and cb.packed_prompts == 3
    cb.run_until_done()
    assert [r.output_ids for r in reqs] == [_expected(p, 4) for p in prompts]
    assert not cb.has_work() and sorted(cb.free_slots) == [0, 1, 2, 3]


def test_long_prompts_are_admitted_in_chunks():
    prompts = [[i + 1, 2 * i + 3, 5] + [7] * (3 * i) for i in range(6)]      # 3 .. 18 tokens
    for on in (True, False):
        eng = _FakeEngine()
        eng.chunks = []
        cb = ContinuousBatcher(eng, max_num_seqs=3, max_seq_len=64, piggyback=on, max_prefill_tokens=4)
        reqs = [cb.submit(SeqRequest(str(i), list(p), max_new_tokens=5)) for i, p in enumerate(prompts)]
        cb.run_until_done()
        assert [r.output_ids for r in reqs] == [_expected(p, 5) for p in prompts]
        assert max(eng.chunks) == 4 and sum(eng.chunks) == sum(len(p) for p in prompts)
        assert not cb.has_work() and sorted(cb.free_slots) == [0, 1, 2]


def test_eos_frees_the_slot_and_callbacks_fire():
    eng = _FakeEngine()
    seen = []
    cb = ContinuousBatcher(eng, max_num_seqs=2, max_seq_len=64, on_token=lambda r, t: seen.append((r.request_id, t)))
    exp = _expected([1, 2, 3], 8)
    a = cb.submit(SeqRequest("a", [1, 2, 3], max_new_tokens=8, eos_token_id=exp[2]))
    b = cb.submit(SeqRequest("b", [4, 4], max_new_tokens=5))
    c = cb.submit(SeqRequest("c", [9], max_new_tokens=4))
    cb.run_until_done()
    assert a.output_ids == exp[:3] and a.finish_reason == "stop"
    assert b.output_ids == _expected([4, 4], 5) and b.finish_reason == "length"
    assert c.output_ids == _expected([9], 4)
    assert sorted(t for r, t in seen if r == "c") == sorted(c.output_ids)
    assert not cb.has_work() and sorted(cb.free_slots) == [0, 1]
    assert cb.generated == 3 + 5 + 4


def test_capacity_is_checked_at_submit():
    cb = ContinuousBatcher(_FakeEngine(), max_num_seqs=1, max_seq_len=8)
    try:
        cb.submit(SeqRequest("x", [1] * 6, max_new_tokens=6))
    except ValueError:
        return
    raise AssertionError("over-capacity request was accepted")


def test_failed_step_releases_requests_and_slots():
    """An exception inside step() must not leave has_work() true for ever: fail_all() ends the admitted requests, frees
    their slots and the batcher serves the next request normally; empty prompts are refused at submit."""
    eng = _FakeEngine()
    cb = ContinuousBatcher(eng, max_num_seqs=2, max_seq_len=32)
    a = cb.submit(SeqRequest("a", [1, 2, 3], max_new_tokens=4))
    b = cb.submit(SeqRequest("b", [4, 5], max_new_tokens=4))
    cb.step()
    cb.step()
    assert cb.running

    def boom(*args, **kw):
        raise RuntimeError("device fault")
    good_decode, eng.decode = eng.decode, boom
    try:
        while cb.has_work():
            cb.step()
        raise AssertionError("the failing step did not raise")
    except RuntimeError:
        failed = cb.fail_all("error")
    assert {r.request_id for r in failed} <= {"a", "b"} and failed
    assert all(r.finished and r.finish_reason == "error" for r in failed)
    assert not cb.has_work() and sorted(cb.free_slots) == [0, 1]
    eng.decode = good_decode
    c = cb.submit(SeqRequest("c", [9, 9], max_new_tokens=3))
    while cb.has_work():
        cb.step()
    assert c.output_ids == _expected([9, 9], 3)
    try:
        cb.submit(SeqRequest("empty", [], max_new_tokens=3))
    except ValueError:
        return
    raise AssertionError("empty prompt was accepted")


class _FakeEnginePackedMixed(_FakeEngine):
    packed_with_decode = True
    packed_mixed = 0

    def prefill_packed(self, prompts, slots, with_decode=False):
        first = super().prefill_packed(prompts, slots)
        if not with_decode:
            return first
        self.packed_mixed += 1
        return (first,) + self.decode(None, 1, chain=True)


def test_packed_prompts_and_decode_rows_share_one_pass():
    """Running sequences + two or more waiting prompts: one pass admits the prompts and advances the running ones;
    tokens are those of separate passes."""
    prompts = [[i + 1, 2 * i + 3, 5] + [7] * i for i in range(11)]
    eng = _FakeEnginePackedMixed()
    cb = ContinuousBatcher(eng, max_num_seqs=6, max_seq_len=64, max_prefill_per_step=4, max_prefill_tokens=32)
    cb.packed_mixed = True                            # (BIGDL_SERVING_PACKED_MIXED=1)
    reqs = [cb.submit(SeqRequest(str(i), list(p), max_new_tokens=2 + i % 5)) for i, p in enumerate(prompts)]
    cb.run_until_done()
    assert eng.packed_mixed > 0 and cb.packed_mixed_steps == eng.packed_mixed
    assert [r.output_ids for r in reqs] == [_expected(p, 2 + i % 5) for i, p in enumerate(prompts)]
    assert not cb.has_work() and sorted(cb.free_slots) == list(range(6))
