"""Engine-level self-speculative decoding (bigdl_amd/models/spec_engine.py): the output must be exactly the target's
own greedy continuation, whatever the draft proposes, and the draft must work on the target's cache."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(qt_draft, seed=5):
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import DecoderConfig, build_random_decoder
    from bigdl_amd.models.spec_engine import SelfSpeculativeEngine, quantized_draft_of
    cfg = DecoderConfig(vocab_size=2048, hidden_size=1024, intermediate_size=2816, num_layers=4, num_heads=8, num_kv_heads=4,
                        head_dim=128, max_position=2048)
    target = build_random_decoder(cfg, Q.FP16, device="cuda", seed=seed)
    draft = quantized_draft_of(target, qt_draft)
    return target, draft, SelfSpeculativeEngine(target, draft, max_step_draft=5)


@pytest.mark.parametrize("qt", ["sym_int4", "sym_int8"])
@pytest.mark.parametrize("kv_fp8", [False, True])
def test_speculative_output_is_the_targets_greedy_continuation(qt, kv_fp8):
    from bigdl_amd import qtypes as Q
    target, draft, spec = _pair(Q.ggml_tensor_qtype[qt])
    ids = torch.randint(3, 2048, (1, 33), generator=torch.Generator().manual_seed(1)).cuda()
    want = target.generate(ids, max_new_tokens=40, kv_fp8=kv_fp8)
    got = spec.generate(ids, max_new_tokens=40, kv_fp8=kv_fp8)
    assert got.shape == want.shape
    # verification rows run the fp16 GEMV at k + 1 rows instead of 1: identical kernels per row, so identical tokens
    assert torch.equal(got, want), (got[0, 33:].tolist(), want[0, 33:].tolist())
    assert draft.cache is target.cache                      # one cache, no draft prefill
    assert spec.n_drafted > 0 and 0 <= spec.n_matched <= spec.n_drafted
    assert sum(spec.accept_num) >= 39
    if qt == "sym_int8":
        assert spec.n_matched / spec.n_drafted > 0.5        # an 8-bit copy of the same weights mostly agrees


def test_eos_stops_the_round():
    from bigdl_amd import qtypes as Q
    target, draft, spec = _pair(Q.SYM_INT8)
    ids = torch.randint(3, 2048, (1, 20), generator=torch.Generator().manual_seed(2)).cuda()
    want = target.generate(ids, max_new_tokens=24)[0, 20:].tolist()
    eos = want[9]
    got = spec.generate(ids, max_new_tokens=24, eos_token_id=eos)[0, 20:].tolist()
    cut = want.index(eos)
    assert got == want[: cut + 1]


@pytest.mark.parametrize("lag", [0, 1])
def test_threshold_stops_the_draft_and_keeps_the_greedy_output(lag):
    """th_stop_draft: drafting ends early when the draft is unsure (a random-init draft almost always is at 0.99, never at
    0.0), the verified text does not change, and the confidence never makes the device wait (lag 1)."""
    from bigdl_amd import qtypes as Q
    target, draft, spec = _pair(Q.SYM_INT4)
    spec.min_step_draft = 2
    ids = torch.randint(3, 2048, (1, 33), generator=torch.Generator().manual_seed(1)).cuda()
    want = target.generate(ids, max_new_tokens=40)
    eager = spec.generate(ids, max_new_tokens=40, th_stop_draft=0.0, stop_lag=lag)           # never stops early
    assert torch.equal(eager, want) and max(spec.draft_num) == 5
    full = spec.n_drafted
    strict = spec.generate(ids, max_new_tokens=40, th_stop_draft=0.999, stop_lag=lag)        # stops as soon as it may
    assert torch.equal(strict, want)
    # min_step_draft tokens, plus the one that was in flight when the host saw the low confidence (lag 1)
    assert max(spec.draft_num[:-1]) <= 2 + lag and spec.n_drafted < full
    moved = spec.generate(ids, max_new_tokens=40, th_stop_draft=0.5, auto_th_stop_draft=True, stop_lag=lag)
    assert torch.equal(moved, want) and spec.th_stop_draft != 0.5                             # the threshold adapted


def test_sampled_speculation_follows_the_target_distribution():
    """Acceptance / resampling on the device: (1) with top_k = 1 both distributions are one-hot, so sampling must
    reproduce the greedy text; (2) the first generated token after a fixed prefix, over many seeds, is distributed as
    the target's own filtered distribution (chi-square on the 8 most likely tokens), although most draws come from
    accepted draft tokens or residual resampling."""
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.spec_engine import _filtered_probs
    target, draft, spec = _pair(Q.SYM_INT8)
    ids = torch.randint(3, 2048, (1, 21), generator=torch.Generator().manual_seed(4)).cuda()
    want = target.generate(ids, max_new_tokens=24)
    got = spec.generate(ids, max_new_tokens=24, do_sample=True, top_k=1)
    assert torch.equal(got, want)
    # distribution of the 2nd new token given the greedy 1st (so that it is produced by the speculative round)
    target.ensure_cache(1, 64)
    target.cache.reset()
    prefix = want[:, :22]
    q = _filtered_probs(target.prefill(prefix)[:, -1, :].float() / 1.0, 0.7, 8, 1.0)[0]
    top = torch.topk(q, 8).indices.tolist()
    counts = {t: 0 for t in top}
    n = 300
    torch.manual_seed(0)
    first = int(want[0, 21])
    used = 0
    for _ in range(n):
        out = spec.generate(ids, max_new_tokens=3, do_sample=True, temperature=0.7, top_k=8)
        if int(out[0, 21]) != first:          # the prompt pass sampled another first token: different condition
            continue
        used += 1
        counts[int(out[0, 22])] = counts.get(int(out[0, 22]), 0) + 1
    assert used > 30 and set(counts) == set(top), "a token outside the target's top-k was produced"
    chi2 = sum((counts[t] - used * float(q[t])) ** 2 / max(used * float(q[t]), 1e-9) for t in top if float(q[t]) * used >= 1)
    assert chi2 < 30.0, (chi2, counts, [round(float(q[t]) * used, 1) for t in top])          # 7 dof: p(chi2 > 30) < 1e-4
    assert spec.n_drafted > 0


def test_oracle_acceptance_sets_the_accepted_length():
    from bigdl_amd import qtypes as Q
    target, draft, spec = _pair(Q.SYM_INT4)
    ids = torch.randint(3, 2048, (1, 20), generator=torch.Generator().manual_seed(2)).cuda()
    spec.generate(ids, max_new_tokens=36, oracle_accept=1.0)
    assert spec.n_matched == spec.n_drafted and max(spec.draft_num) == 5        # every round accepts all, k grows to the cap
    spec.generate(ids, max_new_tokens=36, oracle_accept=0.0)
    assert spec.n_matched == 0


def test_default_kept_layers():
    from bigdl_amd.models.spec_engine import default_kept_layers
    for n, k in ((32, 12), (32, 8), (32, 16), (32, 4), (32, 1), (6, 5), (80, 24), (4, 9)):
        kept = default_kept_layers(n, k)
        assert len(kept) == min(k, n) and kept == sorted(set(kept)) and kept[0] == 0 and kept[-1] <= n - 1
        if 4 <= k < n:
            assert kept[:2] == [0, 1] and kept[-2:] == [n - 2, n - 1]


@pytest.mark.parametrize("kept", [[0, 1, 3], [0, 5], [0, 1, 2, 3, 4, 5]])
@pytest.mark.parametrize("kv_fp8", [False, True])
def test_layer_skipping_draft_shares_the_cache_and_keeps_the_output(kept, kv_fp8):
    """a draft that runs a subset of the (8-bit copy of the) target's layers on the target's own cache rows: the output
    is still the target's greedy continuation; with all layers kept it is the full-depth draft"""
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import DecoderConfig, build_random_decoder
    from bigdl_amd.models.spec_engine import SelfSpeculativeEngine, layer_skip_draft_of, quantized_draft_of
    cfg = DecoderConfig(vocab_size=2048, hidden_size=1024, intermediate_size=2816, num_layers=6, num_heads=8, num_kv_heads=4,
                        head_dim=128, max_position=2048)
    target = build_random_decoder(cfg, Q.FP16, device="cuda", seed=7)
    full = quantized_draft_of(target, Q.SYM_INT8)
    draft = layer_skip_draft_of(full, kept)
    assert draft.cache_layer == kept and len(draft.layers) == len(kept) and draft.layers[-1] is full.layers[kept[-1]]
    spec = SelfSpeculativeEngine(target, draft, max_step_draft=4)
    ids = torch.randint(3, 2048, (1, 29), generator=torch.Generator().manual_seed(2)).cuda()
    got = spec.generate(ids, max_new_tokens=32, kv_fp8=kv_fp8)
    assert got.shape == (1, 29 + 32) and torch.equal(got[:, :29], ids)
    # every generated token is the target's choice at its position: teacher-forced dense fp32 logits over the produced
    # sequence; a token may differ from the dense argmax only by a rounding-sized logit gap (one-row and multi-row fp16
    # kernels sum in different orders, the e5m2 cache rounds keys and values)
    from bigdl_amd.models.decoder import dense_reference_logits
    ref = dense_reference_logits(target, got[:, :-1])[0, -32:]
    new = got[0, -32:]
    gap = ref.max(-1).values - ref.gather(1, new[:, None]).squeeze(1)
    assert bool((gap <= 0.25 * ref.std(-1)).all()) and float((ref.argmax(-1) == new).float().mean()) >= 0.75
    assert draft.cache is target.cache and spec.n_drafted > 0
    if len(kept) == 6:
        ref = SelfSpeculativeEngine(target, full, max_step_draft=4)
        ref.generate(ids, max_new_tokens=32, kv_fp8=kv_fp8)
        assert (spec.n_matched, spec.n_drafted) == (ref.n_matched, ref.n_drafted)
        assert spec.n_matched / spec.n_drafted > 0.5
    with pytest.raises(AssertionError):
        layer_skip_draft_of(full, [3, 1])


def test_search_kept_layers_keeps_the_layers_that_matter():
    """greedy elimination on a 6-layer pair whose layer 3 was made (almost) an identity in the target and the draft: it
    is the first layer the search drops, and the reported acceptance is reproducible"""
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import DecoderConfig, build_random_decoder
    from bigdl_amd.models.spec_engine import (SelfSpeculativeEngine, layer_skip_draft_of, quantized_draft_of,
                                              search_kept_layers)
    cfg = DecoderConfig(vocab_size=2048, hidden_size=1024, intermediate_size=2816, num_layers=6, num_heads=8, num_kv_heads=4,
                        head_dim=128, max_position=2048)
    target = build_random_decoder(cfg, Q.FP16, device="cuda", seed=9)
    for name in ("o", "down"):                      # layer 3 adds (nearly) nothing to the residual stream
        lin = getattr(target.layers[3], name)
        lin.w.mul_(1e-3)
    full = quantized_draft_of(target, Q.SYM_INT8)
    prompts = [torch.randint(3, 2048, (1, 24), generator=torch.Generator().manual_seed(s_)).cuda() for s_ in (1, 2)]
    seen = []
    kept, score = search_kept_layers(target, full, prompts, keep=5, max_new_tokens=24, log=seen.append)
    assert kept == [0, 1, 2, 4, 5] and len(seen) == 1 and "dropped layer 3" in seen[0]
    spec = SelfSpeculativeEngine(target, layer_skip_draft_of(full, kept), max_step_draft=4, adapt=False)
    drafted = matched = 0
    for ids in prompts:
        spec.generate(ids, max_new_tokens=24)
        drafted, matched = drafted + spec.n_drafted, matched + spec.n_matched
    assert abs(matched / drafted - score) < 1e-9 and score > 0.4
    kept2, _ = search_kept_layers(target, full, prompts[:1], keep=3, max_new_tokens=12, candidates_per_round=2)
    assert len(kept2) == 3 and kept2[0] == 0 and kept2[-1] == 5
