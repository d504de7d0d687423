"""Decoding strategies and KV caches on the CPU (HF forward path): parity of lookup / speculative decoding with
plain greedy decoding, cache bookkeeping, SnapKV selection."""
import copy

import pytest
import torch
from transformers import LlamaConfig, LlamaForCausalLM

from bigdl_amd.optimize import optimize_model
from bigdl_amd.transformers import kv
from bigdl_amd.transformers.lookup import PromptLookupCandidateGenerator
from bigdl_amd.transformers.speculative import attach_speculative


@pytest.fixture(scope="module")
def tiny():
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                      num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=512,
                      tie_word_embeddings=False)
    cfg.eos_token_id = None
    m = LlamaForCausalLM(cfg).eval()
    m.generation_config.eos_token_id = None
    return m


def _greedy(model, ids, n):
    cur = ids
    with torch.no_grad():
        for _ in range(n):
            nxt = model(cur).logits[:, -1].argmax(-1, keepdim=True)
            cur = torch.cat([cur, nxt], 1)
    return cur


def test_normal_and_fp8_cache_update_crop():
    torch.manual_seed(1)
    for cls, tol in ((kv.DynamicNormalCache, 0.0), (kv.DynamicFp8Cache, 0.08)):
        c = cls()
        ks, vs = [], []
        for n in (7, 1, 1, 3):
            k, v = torch.randn(2, 2, n, 16), torch.randn(2, 2, n, 16)
            ks.append(k), vs.append(v)
            ko, vo = c.update(k, v, 0)
        ref_k, ref_v = torch.cat(ks, 2), torch.cat(vs, 2)
        assert ko.shape == ref_k.shape and c.get_seq_length() == 12
        assert (ko - ref_k).abs().max() <= tol * ref_k.abs().max() + 1e-6
        assert (vo - ref_v).abs().max() <= tol * ref_v.abs().max() + 1e-6
        c.layers[0].crop(max_length=8)
        assert c.get_seq_length() == 8
        ko, _ = c.update(torch.zeros(2, 2, 1, 16), torch.zeros(2, 2, 1, 16), 0)
        assert ko.shape[2] == 9 and (ko[:, :, :8] - ref_k[:, :, :8]).abs().max() <= tol * ref_k.abs().max() + 1e-6


def test_compress_kv_keeps_attended_positions():
    torch.manual_seed(2)
    b, hq, hkv, s, d, win, cap = 1, 4, 2, 200, 16, 8, 40
    k = torch.randn(b, hkv, s, d) * 0.1
    v = torch.randn(b, hkv, s, d)
    q = torch.randn(b, hq, s, d) * 0.1
    hot = [17, 90, 143]
    # make the window queries attend strongly to the hot keys
    direction = torch.randn(d)
    direction /= direction.norm()
    for p in hot:
        k[:, :, p] = direction * 20
    q[:, :, -win:] = direction * 5
    ck, cv = kv.compress_kv(q, k, v, window_size=win, max_capacity_prompt=cap, kernel_size=5, num_key_value_groups=2)
    assert ck.shape == (b, hkv, cap, d) and cv.shape == (b, hkv, cap, d)
    assert torch.equal(ck[:, :, -win:], k[:, :, -win:])            # the observation window is always kept
    for p in hot:                                                   # hot rows survive, in order
        assert (ck[0, 0] == k[0, 0, p]).all(-1).any()
    short_k, _ = kv.compress_kv(q[:, :, :30], k[:, :, :30], v[:, :, :30], window_size=win, max_capacity_prompt=cap)
    assert short_k.shape[2] == 30


def test_compress_cache_bookkeeping():
    c = kv.DynamicCompressCache(window_size=8, max_capacity_prompt=32, kernel_size=5)
    k, v, q = torch.randn(1, 2, 100, 16), torch.randn(1, 2, 100, 16), torch.randn(1, 4, 100, 16)
    ko, vo = c.update(k, v, 0, query_states=q, num_key_value_groups=2)
    assert ko.shape[2] == 100                     # prefill attends to the whole prompt
    assert c.get_seq_length() == 100 and c.layers[0]._len == 32
    ko, vo = c.update(torch.randn(1, 2, 1, 16), torch.randn(1, 2, 1, 16), 0, query_states=torch.randn(1, 4, 1, 16),
                      num_key_value_groups=2)
    assert ko.shape[2] == 33 and c.get_seq_length() == 101
    assert c.layers[0].get_mask_sizes(1)[0] == 34


def test_lookup_table_matches_bruteforce():
    torch.manual_seed(3)
    toks = torch.randint(0, 6, (1, 80))
    cg = PromptLookupCandidateGenerator(num_output_tokens=4, max_matching_ngram_size=3)
    cg.init_look_up_table(toks[:, :50])
    cg.update_look_up_table(toks[0, 50:])
    seq = toks[0].tolist()
    expect = []
    for n in (3, 2, 1):
        tail = seq[-n:]
        hits = [i for i in range(len(seq) - n) if seq[i:i + n] == tail]
        if hits:
            expect = seq[hits[-1] + n: hits[-1] + n + 4]
            break
    assert cg.get_candidates() == expect
    n0 = cg.num_output_tokens
    cg.update_candidate_strategy(4, 4)
    assert cg.num_output_tokens == n0 + 1
    cg.update_candidate_strategy(5, 1)
    assert cg.num_output_tokens == n0


def test_lookup_generate_equals_greedy(tiny):
    model = optimize_model(copy.deepcopy(tiny), low_bit="sym_int8")
    torch.manual_seed(4)
    ids = torch.randint(0, 320, (1, 12)).repeat(1, 3)       # repetitive prompt so drafts exist
    ref = _greedy(model, ids, 24)
    out = model.generate(ids, max_new_tokens=24, lookahead=3, max_matching_ngram_size=2)
    assert torch.equal(out, ref)
    assert model.n_drafted > 0 and model.n_token_generated == 24
    hf = model.generate(ids, max_new_tokens=24, do_sample=False)
    assert torch.equal(hf[:, : ref.shape[1]], ref)


def test_speculative_generate_equals_target_greedy(tiny):
    target = optimize_model(copy.deepcopy(tiny), low_bit="sym_int8")
    draft = optimize_model(copy.deepcopy(tiny), low_bit="sym_int4")
    attach_speculative(target, draft)
    torch.manual_seed(5)
    ids = torch.randint(0, 320, (1, 9))
    ref = _greedy(target, ids, 20)
    out = target.generate(ids, max_new_tokens=20, max_step_draft=4, th_stop_draft=0.0, min_step_draft=1)
    assert torch.equal(out, ref)
    assert target.n_drafted > 0 and len(target.draft_time) == len(target.verify_time) == len(target.accept_num)
    assert sum(target.accept_num) + 1 >= 20
    # identical draft: every draft token is accepted
    same = optimize_model(copy.deepcopy(tiny), low_bit="sym_int8")
    attach_speculative(target, same)
    out2 = target.generate(ids, max_new_tokens=20, max_step_draft=4, th_stop_draft=0.0, auto_th_stop_draft=False)
    assert torch.equal(out2, ref)
    assert target.n_matched == target.n_drafted
    # sampling path runs and yields valid tokens
    out3 = target.generate(ids, max_new_tokens=12, do_sample=True, top_k=20, temperature=0.8)
    assert out3.shape[1] == ids.shape[1] + 12 and int(out3.max()) < 320


def test_fp8_and_compress_policy_through_model(tiny, monkeypatch):
    model = optimize_model(copy.deepcopy(tiny), low_bit="sym_int8")
    torch.manual_seed(6)
    ids = torch.randint(0, 320, (1, 80))
    ref = model.generate(ids, max_new_tokens=6, do_sample=False)
    monkeypatch.setenv("IPEX_LLM_QUANTIZE_KV_CACHE", "1")
    out = model(ids, use_cache=True)
    assert isinstance(out.past_key_values, kv.DynamicFp8Cache)
    monkeypatch.delenv("IPEX_LLM_QUANTIZE_KV_CACHE")
    out = model(ids, use_cache=True)
    assert isinstance(out.past_key_values, kv.DynamicNormalCache)
    # SnapKV with a capacity above the prompt length is exact
    monkeypatch.setenv("IPEX_LLM_COMPRESS_KV_CACHE", "1")
    out = model(ids, use_cache=True)
    assert isinstance(out.past_key_values, kv.DynamicCompressCache)
    got = model.generate(ids, max_new_tokens=6, do_sample=False)
    assert torch.equal(got, ref)
    # and with real compression the model still decodes (positions keep counting the real length)
    cache = kv.DynamicCompressCache(window_size=8, max_capacity_prompt=32, kernel_size=5)
    o1 = model(ids, past_key_values=cache, use_cache=True)
    assert cache.layers[0]._len == 32 and cache.get_seq_length() == 80
    nxt = o1.logits[:, -1:].argmax(-1)
    o2 = model(nxt, past_key_values=cache, use_cache=True)
    assert o2.logits.shape[1] == 1 and torch.isfinite(o2.logits).all() and cache.get_seq_length() == 81
