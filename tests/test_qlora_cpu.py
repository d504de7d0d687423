"""LoRA / QLoRA / QA-LoRA / ReLoRA / LISA on the CPU path + data-parallel gradient sync (gloo, 2 processes)."""
import copy
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from transformers import LlamaConfig, LlamaForCausalLM


def _tiny():
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                      num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=128,
                      tie_word_embeddings=False)
    return LlamaForCausalLM(cfg)


def _qmodel(low_bit="nf4"):
    from bigdl_amd.optimize import optimize_model
    return optimize_model(_tiny(), low_bit=low_bit, optimize_llm=False, modules_to_not_convert=["lm_head"])


@pytest.mark.parametrize("mode,low_bit", [("qlora", "nf4"), ("qalora", "sym_int4"), ("lora", "bf16")])
def test_lora_training_reduces_loss_and_freezes_base(mode, low_bit, tmp_path):
    from bigdl_amd.transformers.qlora import (LoraConfig, LoraLowBitLinear, PeftModel, get_peft_model,
                                              prepare_model_for_kbit_training)
    model = _qmodel(low_bit)
    model = prepare_model_for_kbit_training(model, use_gradient_checkpointing=False)
    cfg = LoraConfig(r=8, lora_alpha=16, lora_dropout=0.0, target_modules=["q_proj", "k_proj", "v_proj", "o_proj"],
                     bias="none", task_type="CAUSAL_LM", training_mode=mode)
    pm = get_peft_model(model, cfg)
    trainable, total = pm.print_trainable_parameters()
    assert 0 < trainable < 0.2 * total
    if mode != "lora":
        assert isinstance(pm.base_model.model.layers[0].self_attn.q_proj, LoraLowBitLinear)
    base_bytes = pm.base_model.model.layers[0].self_attn.q_proj.base_layer.weight.data.clone().view(torch.uint8)
    torch.manual_seed(1)
    ids = torch.randint(0, 320, (4, 16))
    opt = torch.optim.AdamW([p for p in pm.parameters() if p.requires_grad], lr=5e-3)
    pm.train()
    losses = []
    import contextlib
    # a bf16 base model trains under autocast (what ``TrainingArguments(bf16=True)`` sets up)
    ctx = (lambda: torch.autocast("cpu", dtype=torch.bfloat16)) if mode == "lora" else contextlib.nullcontext
    for _ in range(12):
        with ctx():
            out = pm(input_ids=ids, labels=ids)
        out.loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(out.loss.item())
    assert losses[-1] < losses[0] - 0.2, losses
    assert torch.equal(pm.base_model.model.layers[0].self_attn.q_proj.base_layer.weight.data.view(torch.uint8), base_bytes)
    # adapter-only checkpoint round trip
    pm.eval()
    with torch.no_grad(), ctx():
        ref = pm(input_ids=ids).logits.float()
    pm.save_pretrained(str(tmp_path / "adapter"))
    assert os.path.exists(tmp_path / "adapter" / "adapter_model.safetensors")
    assert os.path.getsize(tmp_path / "adapter" / "adapter_model.safetensors") < 2e6
    fresh = PeftModel.from_pretrained(_qmodel(low_bit), str(tmp_path / "adapter"))
    with torch.no_grad(), ctx():
        got = fresh(input_ids=ids).logits.float()
    tol = 3e-2 if mode == "lora" else 1e-4
    torch.testing.assert_close(got, ref, rtol=tol, atol=tol)
    # merge: dense weights reproduce adapter outputs
    merged = fresh.merge_and_unload()
    with torch.no_grad(), ctx():
        m = merged(input_ids=ids).logits.float()
    tol = 6e-2 if mode == "lora" else 2e-3
    torch.testing.assert_close(m, ref, rtol=tol, atol=tol)


def test_relora_merge_requantizes_and_resets():
    from bigdl_amd.transformers.qlora import LoraConfig, get_peft_model
    from bigdl_amd.transformers.relora import ReLoRAScheduler, merge_and_reinit, reset_optimizer
    pm = get_peft_model(_qmodel("sym_int8"), LoraConfig(r=4, lora_alpha=8, target_modules=["q_proj", "v_proj"],
                                                        training_mode="relora"))
    ids = torch.randint(0, 320, (2, 12))
    opt = torch.optim.AdamW([p for p in pm.parameters() if p.requires_grad], lr=1e-2)
    for _ in range(3):
        pm(input_ids=ids, labels=ids).loss.backward()
        opt.step()
        opt.zero_grad()
    pm.eval()
    with torch.no_grad():
        before = pm(input_ids=ids).logits
    touched = merge_and_reinit(pm)
    assert touched and all(float(m.lora_B["default"].weight.abs().sum()) == 0
                           for m in pm.modules() if hasattr(m, "lora_B"))
    with torch.no_grad():
        after = pm(input_ids=ids).logits          # adapters are zero again; the base absorbed them (int8 rounding)
    assert ((after - before).abs().mean() / before.abs().mean()).item() < 0.02
    reset_optimizer(opt, touched)
    st = opt.state[touched[0]]
    assert (st["exp_avg"] == 0).float().mean() > 0.85
    inner = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    sched = ReLoRAScheduler(opt, inner, relora_steps=10, warmup_steps=4)
    lrs = []
    for _ in range(14):
        opt.step()
        sched.step()
        lrs.append(sched.get_last_lr()[0])
    # the scheduler's constructor performs the initial step: lrs[i] is the rate of step i + 2
    assert lrs[7] == pytest.approx(1e-2) and lrs[8] < 1e-4 and lrs[12] == pytest.approx(1e-2)


def test_lisa_switches_layers():
    from types import SimpleNamespace
    from bigdl_amd.transformers.lisa import DynamicLayerActivationCallback
    model = _tiny()
    cb = DynamicLayerActivationCallback(n_layers=1, interval_steps=2, model=model, seed=0)
    cb.on_step_begin(None, SimpleNamespace(global_step=0), None)
    active = [any(p.requires_grad for p in l.parameters()) for l in model.model.layers]
    assert sum(active) == 1 and len(cb.active_layers_indices) == 1
    assert all(p.requires_grad for p in model.model.embed_tokens.parameters())


def _dp_worker(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                      BIGDL_FORCE_CPU="1")
    torch.set_num_threads(2)
    from bigdl_amd.parallel import init_distributed
    from bigdl_amd.parallel.data_parallel import DataParallelAdapter
    from bigdl_amd.transformers.qlora import LoraConfig, get_peft_model
    init_distributed("gloo")
    torch.manual_seed(100 + rank)                     # different adapter init per rank: broadcast must align them
    pm = get_peft_model(_qmodel("sym_int8"), LoraConfig(r=4, lora_alpha=8, target_modules=["q_proj", "v_proj"]))
    for p in pm.parameters():
        if p.requires_grad:
            p.data.add_(0.01 * torch.randn_like(p))
    opt = torch.optim.SGD([p for p in pm.parameters() if p.requires_grad], lr=0.1)
    dp = DataParallelAdapter(pm, opt)
    torch.manual_seed(7)
    data = torch.randint(0, 320, (4, 10))
    mine = data[rank * 2:(rank + 1) * 2]
    # reference: same start, full batch on one process
    ref = copy.deepcopy(pm)
    ref_opt = torch.optim.SGD([p for p in ref.parameters() if p.requires_grad], lr=0.1)
    ref(input_ids=data, labels=data).loss.backward()
    ref_opt.step()
    dp.step(pm(input_ids=mine, labels=mine).loss)
    for (n, a), (_, b) in zip(pm.named_parameters(), ref.named_parameters()):
        if a.requires_grad:
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5, msg=n)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_matches_full_batch():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(port,), nprocs=2, join=True)


def test_training_forward_keeps_all_positions_through_a_converted_lm_head():
    """The last-token lm_head shortcut is an inference optimisation: a training step with labels on a sequence longer
    than 10 tokens needs logits for every position."""
    import torch
    from transformers import LlamaConfig, LlamaForCausalLM
    from bigdl_amd import optimize_model
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=4,
                      num_key_value_heads=4, max_position_embeddings=64)
    m = optimize_model(LlamaForCausalLM(cfg), low_bit="bf16")
    ids = torch.randint(0, 64, (2, 14))
    m.eval()
    with torch.no_grad():
        assert m(input_ids=ids).logits.shape[1] == 1
    m.train()
    for p in m.parameters():
        p.requires_grad_(p.is_floating_point())
    out = m(input_ids=ids, labels=ids)
    assert out.logits.shape[1] == 14 and torch.isfinite(out.loss)
