"""Native decoder engine on the GPU: graph decode == eager decode == dense fp32 reference."""
import math

import pytest
import torch

from bigdl_amd import qtypes as Q

pytestmark = pytest.mark.gpu


def _tiny(**kw):
    from bigdl_amd.models.decoder import DecoderConfig
    base = dict(vocab_size=320, hidden_size=256, intermediate_size=384, num_layers=2, num_heads=4, num_kv_heads=2,
                head_dim=64, max_position=512)
    base.update(kw)
    return DecoderConfig(**base)


def _dense_reference(model, ids):
    """fp32 forward over the dequantized weights: logits for every position."""
    from bigdl_amd.models.decoder import dense_reference_logits
    return dense_reference_logits(model, ids)


@pytest.mark.parametrize("qt", ["sym_int4", "asym_int4", "fp8_e5m2", "nf4", "fp6"])
def test_prefill_and_decode_match_dense_reference(qt):
    from bigdl_amd.models.decoder import build_random_decoder
    torch.manual_seed(0)
    model = build_random_decoder(_tiny(), Q.ggml_tensor_qtype[qt], seed=3)
    ids = torch.randint(0, 320, (2, 21)).cuda()
    ref = _dense_reference(model, ids)
    model.ensure_cache(2, 64)
    model.cache.reset()
    logits = model.prefill(ids[:, :17], last_only=False)
    torch.testing.assert_close(logits, ref[:, :17], rtol=3e-2, atol=3e-2)
    # token-by-token decode through the graph
    for t in range(17, 21):
        lg, _ = model.decode(ids[:, t:t + 1].contiguous(), 1)
        torch.testing.assert_close(lg.view(2, -1), ref[:, t], rtol=3e-2, atol=3e-2)
    # multi-token decode step (speculative verification shape) after cropping the cache
    model.cache.crop(17)
    lg, _ = model.decode(ids[:, 17:21].contiguous(), 4)
    torch.testing.assert_close(lg.view(2, 4, -1), ref[:, 17:21], rtol=3e-2, atol=3e-2)


def test_graph_equals_eager_and_fp8_kv_close():
    from bigdl_amd.models.decoder import build_random_decoder
    model = build_random_decoder(_tiny(num_layers=3), Q.SYM_INT4, seed=5)
    ids = torch.randint(0, 320, (1, 33)).cuda()
    model.use_graph = True
    a = model.generate(ids, max_new_tokens=12)
    model.use_graph = False
    model._graphs.clear()
    b = model.generate(ids, max_new_tokens=12)
    assert torch.equal(a, b)
    assert model.first_cost > 0 and model.rest_cost_mean > 0 and len(model.last_token_time) == 11
    model.use_graph = True
    model._graphs.clear()
    c = model.generate(ids, max_new_tokens=12, kv_fp8=True)
    # e5m2 KV perturbs logits slightly; the first tokens of a greedy continuation normally agree
    assert c.shape == a.shape and torch.equal(c[:, :34], a[:, :34])
    # the fp8 cache written by prefill is what decode reads: logits of the next step stay close to fp16-KV logits
    lg = {}
    for fp8 in (False, True):
        model.ensure_cache(1, 64, fp8=fp8)
        model.cache.reset()
        model.prefill(ids)
        lg[fp8] = model.decode(a[:, 33:34].contiguous(), 1)[0].clone()
    rel = ((lg[True] - lg[False]).abs().mean() / lg[False].abs().mean()).item()
    assert rel < 0.1, rel


def test_sampling_and_eos():
    from bigdl_amd.models.decoder import build_random_decoder
    model = build_random_decoder(_tiny(), Q.SYM_INT4, seed=7)
    ids = torch.randint(0, 320, (2, 9)).cuda()
    torch.manual_seed(1)
    out = model.generate(ids, max_new_tokens=6, do_sample=True, temperature=0.8, top_k=20, top_p=0.9,
                         repetition_penalty=1.1)
    assert out.shape == (2, 15)
    greedy = model.generate(ids, max_new_tokens=6)
    eos = int(greedy[0, 10])
    out = model.generate(ids[:1], max_new_tokens=6, eos_token_id=eos)
    assert out.shape[1] <= 11 + 1 and int(out[0, -1]) == eos


def test_continuous_batching_matches_per_request_generation():
    """Requests of different lengths share one slot cache: each one's tokens equal its own batch-1 generation."""
    from bigdl_amd.models.decoder import build_random_decoder
    from bigdl_amd.serving.continuous import generate_batch
    torch.manual_seed(0)
    model = build_random_decoder(_tiny(num_layers=3), Q.SYM_INT4, device="cuda", seed=5)
    g = torch.Generator().manual_seed(1)
    prompts = [torch.randint(3, 320, (n,), generator=g).tolist() for n in (5, 33, 17, 64, 9)]
    want = []
    for p in prompts:
        out = model.generate(torch.tensor([p]).cuda(), max_new_tokens=12, do_sample=False)
        want.append(out[0, len(p):].tolist())
    got = generate_batch(model, prompts, max_new_tokens=12, max_num_seqs=3)
    agree = sum(int(a == b) for w, o in zip(want, got) for a, b in zip(w, o))
    total = sum(len(w) for w in want)
    # batch rows use the multi-row GEMV path, batch-1 the single-row one: identical up to fp16 summation order, so
    # a near-tie argmax can flip once and change that request's tail
    assert all(w[0] == o[0] for w, o in zip(want, got)), (want, got)
    assert agree >= 0.8 * total, (want, got)


def test_chunked_prompt_admission_matches_whole_prompt_admission():
    """Prompts admitted 64 tokens per step (with the decode rows of the running slots on the same passes) end in the
    same cache contents and first tokens as prompts admitted whole."""
    from bigdl_amd.models import decoder as D
    from bigdl_amd.serving.continuous import ContinuousBatcher, SeqRequest
    cfg = D.DecoderConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_layers=3, num_heads=4, num_kv_heads=2,
                          head_dim=64, max_position=1024)
    eng = D.build_random_decoder(cfg, device="cuda", seed=3)
    g = torch.Generator().manual_seed(0)
    prompts = [torch.randint(3, 512, (n,), generator=g).tolist() for n in (200, 30, 333)]
    res = {}
    for chunk in (None, 64):
        cb = ContinuousBatcher(eng, max_num_seqs=2, max_seq_len=512, max_prefill_tokens=chunk)
        reqs = [cb.submit(SeqRequest(str(i), p, max_new_tokens=40 if i == 0 else 5)) for i, p in enumerate(prompts)]
        cb.run_until_done()
        res[chunk] = ([r.output_ids for r in reqs], cb.steps)
        assert all(r.finish_reason == "length" for r in reqs)
    assert res[64][1] > res[None][1]                          # more, shorter steps
    whole, chunked = res[None][0], res[64][0]
    assert [w[0] for w in whole] == [c[0] for c in chunked], (whole, chunked)
    agree = sum(int(a == b) for w, c in zip(whole, chunked) for a, b in zip(w, c))
    assert agree >= 0.8 * sum(len(w) for w in whole), (whole, chunked)      # fp16 summation order may flip a near-tie


@pytest.mark.parametrize("mode", [False, True, "int4"])
def test_packed_prompt_pass_matches_one_prompt_per_pass(mode):
    """prefill_packed: three prompts of different lengths in one pass give the logits and cache rows of three passes."""
    from bigdl_amd.models import decoder as D
    cfg = D.DecoderConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_layers=3, num_heads=4, num_kv_heads=2,
                          head_dim=128 if mode == "int4" else 64, max_position=1024)
    eng = D.build_random_decoder(cfg, device="cuda", seed=3)
    g = torch.Generator().manual_seed(0)
    prompts = [torch.randint(3, 512, (n,), generator=g).tolist() for n in (70, 9, 200)]
    slots = [2, 0, 3]
    cache = eng.ensure_cache(4, 512, fp8=mode)
    cache.reset()
    cache.enable_slots()
    want = torch.cat([eng.prefill(torch.tensor([p]).cuda(), slot=s)[:, -1] for p, s in zip(prompts, slots)])
    k_ref, lens_ref = cache.k[1].clone(), list(cache.slot_len)
    cache.reset()
    cache.enable_slots()
    for t in cache.k + cache.v:
        t.zero_()
    got = eng.prefill_packed(prompts, slots)
    rel = ((got - want).abs().mean() / want.abs().mean()).item()
    assert rel < 5e-3, rel
    assert list(cache.slot_len) == lens_ref == [9, 0, 70, 200] and cache.past_len.tolist() == lens_ref
    if mode:          # quantized codes: a rounding flips here and there (the row count picks another GEMM kernel)
        assert (cache.k[1] != k_ref).float().mean().item() < 0.03
    else:
        torch.testing.assert_close(cache.k[1].float(), k_ref.float(), rtol=2e-2, atol=2e-2)


def test_int4_kv_cache_generation_tracks_fp16_cache():
    """Engine on the int4 cache: prompt pass, decode steps, a second prompt pass over the non-empty cache and slot
    mode all run; logits stay close to the fp16-cache engine and the cache takes a quarter of the memory + scales."""
    from bigdl_amd.models import decoder as D
    from bigdl_amd.serving.continuous import generate_batch
    cfg = D.DecoderConfig(vocab_size=512, hidden_size=512, intermediate_size=512, num_layers=3, num_heads=4, num_kv_heads=2,
                          head_dim=128, max_position=1024)
    eng = D.build_random_decoder(cfg, device="cuda", seed=3)
    torch.manual_seed(0)
    ids = torch.randint(3, 512, (2, 150), device="cuda")
    more = torch.randint(3, 512, (2, 20), device="cuda")
    lg, size = {}, {}
    for mode in (False, "int4"):
        cache = eng.ensure_cache(2, 400, fp8=mode)
        cache.reset()
        assert cache.int4 == (mode == "int4")
        size[mode] = cache.nbytes()
        a = eng.prefill(ids)[:, -1].clone()
        tok = a.argmax(-1).view(2, 1)
        b = eng.decode(tok, 1)[0].clone()
        c = eng.decode(tok, 1)[0].clone()
        d = eng.prefill(more)[:, -1].clone()               # prompt pass over a non-empty cache (dequantized read)
        lg[mode] = (a, b, c, d)
        assert cache.host_len == 172
    assert size["int4"] * 32 == size[False] * 9              # 4 bits + one fp16 scale per 32 values = 4.5 of 16 bits
    for x, y in zip(lg[False], lg["int4"]):
        rel = ((x - y).abs().mean() / x.abs().mean()).item()
        assert rel < 0.12, rel
    out = eng.generate(ids, max_new_tokens=8, kv_fp8="int4")
    assert out.shape == (2, 158) and eng.cache.int4
    prompts = [ids[0, :n].tolist() for n in (40, 90, 17)]
    got = generate_batch(eng, prompts, max_new_tokens=6, max_num_seqs=2, kv_fp8="int4")
    assert [len(g) for g in got] == [6, 6, 6]
    want = eng.generate(ids[:1, :40], max_new_tokens=1, kv_fp8="int4")
    assert got[0][0] == int(want[0, -1])


@pytest.mark.parametrize("fp8", [False, True])
def test_mixed_prompt_and_decode_pass_matches_separate_passes(fp8):
    """prefill(with_decode=True) = one decode step of the running slots + the prompt pass of a new one: same logits,
    same cache contents and counters as the two separate calls from the same state."""
    from bigdl_amd.models import decoder as D
    cfg = D.DecoderConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_layers=3, num_heads=4, num_kv_heads=2,
                          head_dim=64, max_position=1024)
    eng = D.build_random_decoder(cfg, device="cuda", seed=3)
    g = torch.Generator().manual_seed(0)
    p0, p1, new = (torch.randint(3, 512, (1, n), generator=g).cuda() for n in (40, 19, 300))
    res = {}
    for mixed in (False, True):
        cache = eng.ensure_cache(4, 512, fp8=fp8)
        cache.reset()
        cache.enable_slots()
        tokens = eng._get_graph(4, 1, True)[1]["tokens"]
        for slot, p in ((0, p0), (2, p1)):
            tokens[slot, 0] = eng.prefill(p, slot=slot)[0, -1].argmax()
            cache.set_active(slot, True)
        eng.decode(None, 1, chain=True)                       # one ordinary step first
        if mixed:
            first, lg, nxt = eng.prefill(new, slot=1, with_decode=True)
        else:
            lg, nxt = eng.decode(None, 1, chain=True)
            lg, nxt = lg.clone(), nxt.clone()
            first = eng.prefill(new, slot=1)
        res[mixed] = (first.clone(), lg[[0, 2]].clone(), nxt[[0, 2]].clone(), cache.past_len.clone(), list(cache.slot_len),
                      cache.k[1][[0, 2], :, :45].float().clone(), cache.k[1][1, :, :300].float().clone())
    a, b = res[False], res[True]
    tol = 0.06 if fp8 else 1e-2
    for i in (0, 1):
        rel = ((a[i] - b[i]).abs().mean() / a[i].abs().mean()).item()
        assert rel < tol, (i, rel)
    assert torch.equal(a[3], b[3]) and a[4] == b[4] == [42, 300, 21, 0]
    for i in (5, 6):
        if fp8:                                               # e5m2 codes: a rounding flip here and there
            assert (a[i] != b[i]).float().mean().item() < 0.03
        else:
            torch.testing.assert_close(a[i], b[i], rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("fp8", [False, True])
def test_context_parallel_prefill_matches_plain_prefill(fp8):
    """One rank, two zigzag chunks: chunked RoPE positions, cache writes at chunk offsets and per-chunk causal
    attention reproduce the plain prompt pass; decoding continues from the cache it leaves."""
    from bigdl_amd.models import decoder as D
    cfg = D.DecoderConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_layers=3, num_heads=4, num_kv_heads=2,
                          head_dim=64, max_position=1024)
    eng = D.build_random_decoder(cfg, device="cuda")
    torch.manual_seed(0)
    ids = torch.randint(0, 512, (1, 600), device="cuda")
    eng.ensure_cache(1, 700, fp8=fp8)
    eng.cache.reset()
    want = eng.prefill(ids)
    _, nxt = eng.decode(want[:, -1].argmax(-1).view(1, 1))
    k_ref = eng.cache.k[1][:, :, :600].clone()
    eng.cache.reset()
    got = eng.prefill_context_parallel(ids)
    rel = ((got - want).abs().mean() / want.abs().mean()).item()
    assert rel < (0.05 if fp8 else 5e-3), rel
    if not fp8:
        torch.testing.assert_close(eng.cache.k[1][:, :, :600].float(), k_ref.float(), rtol=2e-2, atol=2e-2)
    assert eng.cache.host_len == 600 and int(eng.cache.past_len[0]) == 600
    _, nxt2 = eng.decode(got[:, -1].argmax(-1).view(1, 1))
    assert got[:, -1].argmax(-1).item() != want[:, -1].argmax(-1).item() or torch.equal(nxt, nxt2)


def test_engine_snapkv_compression_matches_module_path_selection():
    """A compressed prompt on the engine: the cache holds max_capacity_prompt keys, decode positions keep counting from
    the real prompt length, and the tokens equal an uncompressed engine run on a prompt whose attention is concentrated
    enough that the dropped keys do not matter (here: compare against the torch reference of the same selection)."""
    import torch
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import DecoderConfig, build_random_decoder
    from bigdl_amd.transformers.kv import snapkv_indices
    cfg = DecoderConfig(vocab_size=1024, hidden_size=512, intermediate_size=1024, num_layers=2, num_heads=8, num_kv_heads=4,
                        head_dim=64, max_position=4096)
    eng = build_random_decoder(cfg, Q.SYM_INT8, device="cuda", seed=3)
    ids = torch.randint(3, 1024, (1, 300), generator=torch.Generator().manual_seed(4)).cuda()
    eng.compress_kv = dict(window_size=16, max_capacity_prompt=128, kernel_size=5, pooling="maxpool")
    out_c = eng.generate(ids, max_new_tokens=8)
    assert eng.cache.host_len == 128 + 7 and eng._rope_shift == 300 - 128
    # reference: full-precision replay of what the compressed cache must contain, layer 0: K rows = voted prefix + window
    eng.compress_kv = None
    eng.ensure_cache(1, 512)
    eng.cache.reset()
    eng.prefill(ids)
    k_full = eng.cache.k[0][0, :, :300].clone()                    # [Hkv, 300, D] rotated keys of the prompt
    # recompute layer-0 queries to vote with: run the compressed prefill again and compare its cache to a gather of
    # the full cache by the indices the vote yields from the same q / k
    eng.compress_kv = dict(window_size=16, max_capacity_prompt=128, kernel_size=5, pooling="maxpool")
    eng.cache.reset()
    eng.prefill(ids)
    k_comp = eng.cache.k[0][0, :, :128]
    assert torch.equal(k_comp[:, 112:], k_full[:, 284:300])        # the window is kept verbatim
    # every kept prefix row is one of the original rows, in ascending position order
    d = (k_comp[:, :112, None, :].float() - k_full[:, None, :284, :].float()).abs().amax(-1)      # [Hkv,112,284]
    pos = d.argmin(-1)
    assert float(d.amin(-1).max()) == 0.0 and bool((pos[:, 1:] > pos[:, :-1]).all())
    eng.compress_kv = None
    out_u = eng.generate(ids, max_new_tokens=8)
    assert out_c.shape == out_u.shape


def test_fp8_prompt_pass_is_opt_in_and_close_to_fp16():
    """``prefill_fp8``: prompt GEMMs above the fused-GEMM row limit on hipBLASLt's fp8 path (row-wise e4m3 for activations
    and weights).  Off by default; when on, the logits stay within quantization noise of the fp16 prompt pass and decode
    (which never uses it) continues from the cache it filled."""
    import torch
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import DecoderConfig, build_random_decoder
    cfg = DecoderConfig(vocab_size=512, hidden_size=512, intermediate_size=1408, num_layers=3, num_heads=8, num_kv_heads=4,
                        head_dim=64, max_position=1024)
    eng = build_random_decoder(cfg, Q.SYM_INT4, device="cuda", seed=5)
    assert eng.prefill_fp8 is False
    ids = torch.randint(3, 512, (1, 400)).cuda()
    eng.ensure_cache(1, 512)
    eng.cache.reset()
    ref = eng.prefill(ids, last_only=False).float().clone()
    eng.prefill_fp8 = True
    eng.cache.reset()
    out = eng.prefill(ids, last_only=False).float().clone()
    rel = ((out - ref).abs().mean() / ref.abs().mean()).item()
    assert 0 < rel < 0.15, rel
    assert (out.argmax(-1) == ref.argmax(-1)).float().mean().item() > 0.8
    lg, nxt = eng.decode(out[:, -1].argmax(-1).view(1, 1))
    assert torch.isfinite(lg).all()


@pytest.mark.parametrize("fp8", [False, True])
def test_packed_prompts_with_decode_rows_match_separate_passes(fp8):
    """prefill_packed(with_decode=True) = one decode step of the running slots + the packed prompt pass of two new ones."""
    from bigdl_amd.models import decoder as D
    cfg = D.DecoderConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_layers=3, num_heads=4, num_kv_heads=2,
                          head_dim=64, max_position=1024)
    eng = D.build_random_decoder(cfg, device="cuda", seed=3)
    g = torch.Generator().manual_seed(0)
    p0, p1 = (torch.randint(3, 512, (1, n), generator=g).cuda() for n in (40, 19))
    new = [torch.randint(3, 512, (n,), generator=g).tolist() for n in (120, 33)]
    res = {}
    for mixed in (False, True):
        cache = eng.ensure_cache(5, 512, fp8=fp8)
        cache.reset()
        cache.enable_slots()
        tokens = eng._get_graph(5, 1, True)[1]["tokens"]
        for slot, p in ((0, p0), (2, p1)):
            tokens[slot, 0] = eng.prefill(p, slot=slot)[0, -1].argmax()
            cache.set_active(slot, True)
        eng.decode(None, 1, chain=True)
        if mixed:
            first, lg, nxt = eng.prefill_packed(new, [4, 1], with_decode=True)
        else:
            lg, nxt = eng.decode(None, 1, chain=True)
            lg, nxt = lg.clone(), nxt.clone()
            first = eng.prefill_packed(new, [4, 1])
        res[mixed] = (first.clone(), lg[[0, 2]].clone(), cache.past_len.clone(), list(cache.slot_len),
                      cache.k[1][[0, 2], :, :45].float().clone(), cache.k[1][4, :, :120].float().clone())
    a, b = res[False], res[True]
    tol = 0.06 if fp8 else 1e-2
    for i in (0, 1):
        rel = ((a[i] - b[i]).abs().mean() / a[i].abs().mean()).item()
        assert rel < tol, (i, rel)
    assert torch.equal(a[2], b[2]) and a[3] == b[3] == [42, 33, 21, 0, 120]
    for i in (4, 5):
        if fp8:
            assert (a[i] != b[i]).float().mean().item() < 0.03
        else:
            torch.testing.assert_close(a[i], b[i], rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("start", [0, 4])
def test_attention_sink_eviction_re_rotates_the_kept_keys(start):
    """evict_middle: sinks + recent window kept, keys re-indexed to their positions inside the cache.  Layer 0's keys
    only depend on the embeddings, so the expected cache rows are the unrotated keys rotated (HF's half-rotation
    formula, written out here) at the NEW positions; V^T columns move unchanged; decoding goes on from the new length."""
    from bigdl_amd.models.decoder import build_random_decoder
    eng = build_random_decoder(_tiny(), Q.SYM_INT8, device="cuda", seed=11)
    cfg = eng.cfg
    n, recent = 96, 40
    ids = torch.randint(3, 320, (1, n), device="cuda")
    eng.ensure_cache(1, 256, fp8=False)
    eng.cache.reset()
    eng.prefill(ids)
    lw = eng.layers[0]
    d, hq, hkv = cfg.head_dim, cfg.num_heads, cfg.num_kv_heads
    x = eng._norm(eng.embed[ids[0]].to(torch.float16) * cfg.embed_scale, lw.input_norm)
    qkv = eng._linear_dense(x, lw.qkv).float()
    k_raw = qkv[:, hq * d:(hq + hkv) * d].view(n, hkv, d).transpose(0, 1)              # [hkv, n, d], no positions yet
    inv = eng.inv_freq.float()

    def rope(k, pos):
        ang = pos.float()[:, None] * inv[None]
        cos, sin = ang.cos(), ang.sin()
        k0, k1 = k[..., : d // 2], k[..., d // 2:]
        return torch.cat([k0 * cos - k1 * sin, k1 * cos + k0 * sin], -1)

    before_k = eng.cache.k[0][0, :, :n].float().clone()
    before_v = eng.cache.v[0][0, :, :, :n].clone()
    assert torch.allclose(before_k, rope(k_raw, torch.arange(n, device="cuda")), atol=2e-2)      # the oracle's convention
    drop = eng.evict_middle(start_size=start, recent_size=recent)
    assert drop == n - start - recent and eng.cache.host_len == start + recent
    assert int(eng.cache.past_len[0]) == start + recent
    kept = torch.cat([torch.arange(start), torch.arange(n - recent, n)]).cuda()
    want_k = rope(k_raw[:, kept], torch.arange(start + recent, device="cuda"))
    got_k = eng.cache.k[0][0, :, : start + recent].float()
    assert torch.allclose(got_k, want_k, atol=3e-2)
    assert torch.equal(eng.cache.v[0][0, :, :, : start + recent], before_v[:, :, kept])
    tok = torch.randint(3, 320, (1, 1), device="cuda")
    logits, _ = eng.decode(tok)
    assert torch.isfinite(logits).all() and eng.cache.host_len == start + recent + 1
    assert eng.evict_middle(start_size=start, recent_size=recent + 8) == 0                 # fits: nothing to drop


@pytest.mark.parametrize("kv_fp8", [False, True])
def test_short_prompts_replay_a_captured_prompt_pass(monkeypatch, kv_fp8):
    """prompts of up to 128 tokens are padded to a length bucket and run as one captured pass: same tokens as the eager
    pass, for lengths at, inside and across bucket borders, with buckets reused between calls"""
    import bigdl_amd.models.decoder as D
    from bigdl_amd import qtypes as Q
    cfg = D.DecoderConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, num_kv_heads=2,
                          head_dim=64, max_position=1024)
    model = D.build_random_decoder(cfg, Q.SYM_INT4, device="cuda", seed=3)
    gen = torch.Generator().manual_seed(5)
    for s in (128, 5, 16, 17, 40, 33, 100):          # (the longest first: a cache that has to grow drops its captured passes)
        ids = torch.randint(3, 512, (1, s), generator=gen).cuda()
        monkeypatch.setenv("BIGDL_PREFILL_GRAPH", "0")
        want = model.generate(ids, max_new_tokens=12, kv_fp8=kv_fp8)
        monkeypatch.setenv("BIGDL_PREFILL_GRAPH", "1")
        got = model.generate(ids, max_new_tokens=12, kv_fp8=kv_fp8)
        assert torch.equal(got, want), (s, got[0, s:].tolist(), want[0, s:].tolist())
    assert sorted(model._prefill_graphs) == [16, 32, 48, 128]
    # a longer prompt takes the eager pass; a batch of two as well
    ids = torch.randint(3, 512, (1, 130), generator=gen).cuda()
    model.generate(ids, max_new_tokens=4, kv_fp8=kv_fp8)
    assert sorted(model._prefill_graphs) == [16, 32, 48, 128]
    model.generate(torch.randint(3, 512, (2, 20), generator=gen).cuda(), max_new_tokens=4, kv_fp8=kv_fp8)
    assert not model._prefill_graphs          # (a cache of another shape: captured passes of the old one are dropped)
