"""Host-side pieces of the GPU features that need no GPU to check: layer selection of layer-skipping drafts, argument
checks of the captured training step, the operand predicates of the HIP wrappers, the EAGLE example's speed summary,
the dense-weight-cache policy on a host model."""
import importlib.util
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_kept_layers_shapes():
    from bigdl_amd.models.spec_engine import default_kept_layers
    for n, k in ((32, 12), (32, 8), (32, 16), (32, 4), (32, 1), (6, 5), (80, 24), (4, 9), (28, 14)):
        kept = default_kept_layers(n, k)
        assert len(kept) == min(k, n) and kept == sorted(set(kept)) and 0 <= kept[0] and kept[-1] <= n - 1
        if 4 <= k < n:
            assert kept[:2] == [0, 1] and kept[-2:] == [n - 2, n - 1]
    assert default_kept_layers(32, 8) == [0, 1, 5, 12, 19, 26, 30, 31]


def test_captured_step_rejects_host_batches_and_plain_adam():
    from bigdl_amd.transformers.graph_train import GraphedTrainStep, capturable_adamw
    lin = torch.nn.Linear(4, 4)
    with pytest.raises(ValueError, match="GPU"):
        GraphedTrainStep(lin, torch.optim.SGD(lin.parameters(), lr=0.1), torch.zeros(1, 4, dtype=torch.long))
    opt = capturable_adamw(lin.parameters(), lr=1e-3)          # host parameters: the foreach form, still capturable
    assert all(g["capturable"] for g in opt.param_groups)


def test_wrapper_predicates_say_no_on_the_host():
    from bigdl_amd.ops import hip
    x = torch.zeros(4, 64, dtype=torch.bfloat16)
    assert not hip.lora_ops_ok(x, 8)
    h = torch.zeros(2, 64, dtype=torch.float16)
    assert not hip.moe_norm_route_ok(h, torch.ones(64, dtype=torch.float16), torch.zeros(8, 64, dtype=torch.float16), 8)
    assert hip.gemv_lowbit_moe_sum_ok(2, 14336) and not hip.gemv_lowbit_moe_sum_ok(2, 1400) \
        and not hip.gemv_lowbit_moe_sum_ok(9, 128) and not hip.gemv_lowbit_moe_sum_ok(8, 14336)


def test_eagle_speed_summary(tmp_path):
    spec = importlib.util.spec_from_file_location(
        "eagle_speed", os.path.join(ROOT, "examples", "GPU", "Speculative-Decoding", "EAGLE", "speed.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    p = tmp_path / "a.jsonl"
    p.write_text("\n".join(json.dumps({"question_id": i, "choices": [{"turns": ["x"], "wall_time": [0.5, 0.5],
                                                                         "new_tokens": [40 + 10 * i, 60]}]}) for i in range(2)))
    out = mod.summarize(str(p))
    assert out["questions"] == 2 and out["total_tokens"] == 210 and abs(out["total_time_s"] - 2.0) < 1e-9
    assert abs(out["mean_tokens_per_s"] - 105.0) < 1e-9


def test_dense_cache_policy_leaves_host_models_alone(monkeypatch):
    import bigdl_amd.transformers.low_bit_linear as L
    from transformers import LlamaConfig, LlamaForCausalLM
    from bigdl_amd.optimize import optimize_model
    cfg = LlamaConfig(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2,
                      num_key_value_heads=2)
    model = optimize_model(LlamaForCausalLM(cfg), low_bit="sym_int4")
    monkeypatch.delenv("BIGDL_TRAIN_DENSE_CACHE", raising=False)
    before = L.train_dense_cache_enabled()
    assert L.auto_train_dense_cache(model) == before             # nothing on a GPU: the setting stays as it was
    monkeypatch.setenv("BIGDL_TRAIN_DENSE_CACHE", "1")
    assert L.auto_train_dense_cache(model) is True
    monkeypatch.setenv("BIGDL_TRAIN_DENSE_CACHE", "0")
    assert L.auto_train_dense_cache(model) is False
    monkeypatch.setattr(L, "_TRAIN_DENSE_CACHE", before)


def test_attn_decode_splits_rule(monkeypatch):
    from bigdl_amd.models.decoder import attn_decode_splits
    monkeypatch.delenv("BIGDL_ATTN_SPLITS", raising=False)
    # a lone sequence is cut to fill the chip, ranges of at least 256 keys
    assert attn_decode_splits(1, 32, 1280) == 5
    assert attn_decode_splits(1, 32, 8192) == 8
    assert attn_decode_splits(1, 8, 8192) == 16
    # from 64 (sequence, head) pairs on a range holds at least 1024 keys
    assert attn_decode_splits(2, 32, 1280) == 1
    assert attn_decode_splits(2, 32, 4352) == 4
    assert attn_decode_splits(5, 32, 4352) == 1
    assert attn_decode_splits(8, 8, 4352) == 4
    assert attn_decode_splits(64, 32, 65536) == 1
    monkeypatch.setenv("BIGDL_ATTN_SPLITS", "3")
    assert attn_decode_splits(8, 32, 4352) == 3
    assert attn_decode_splits(8, 32, 300) == 1
