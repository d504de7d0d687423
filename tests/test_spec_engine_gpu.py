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
