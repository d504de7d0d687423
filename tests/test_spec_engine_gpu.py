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
