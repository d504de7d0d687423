"""A QLoRA training step replayed as one hipGraph gives the losses and adapter weights of the eager steps."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmark"))


def _model(seed=0):
    from transformers import LlamaConfig
    from bigdl_amd import qtypes as Q
    from bigdl_amd.transformers.qlora import LoraConfig, get_peft_model, prepare_model_for_kbit_training
    from qlora_bench import random_lowbit_llama
    cfg = LlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=4, max_position_embeddings=256)
    cfg._attn_implementation = "sdpa"
    torch.manual_seed(seed)
    model = random_lowbit_llama(cfg, Q.NF4, torch.device("cuda"), seed=seed)
    model.train()
    model = prepare_model_for_kbit_training(model, use_gradient_checkpointing=False)
    model = get_peft_model(model, LoraConfig(r=8, lora_alpha=16, target_modules=["q_proj", "v_proj", "down_proj"],
                                             lora_dropout=0.0, bias="none", task_type="CAUSAL_LM", training_mode="qlora"))
    return model


@pytest.mark.parametrize("kind", ["adamw", "sgd"])
def test_graphed_step_matches_eager(kind):
    from bigdl_amd.transformers.graph_train import GraphedTrainStep, capturable_adamw
    make = (lambda ps: capturable_adamw(ps, lr=1e-3)) if kind == "adamw" else (lambda ps: torch.optim.SGD(ps, lr=0.05))
    gen = torch.Generator().manual_seed(3)
    two = [torch.randint(3, 512, (2, 48), generator=gen).cuda() for _ in range(2)]
    batches = [two[i % 2] for i in range(8)]      # alternating, so that a replay has to take its batch from the call
    eager = _model()
    params = [p for p in eager.parameters() if p.requires_grad]
    start = [p.detach().clone() for p in params]
    opt = make(params)
    eager_loss = []
    for ids in batches:
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = eager(input_ids=ids, labels=ids).loss
        loss.backward()
        opt.step()
        eager_loss.append(float(loss.detach()))
    graphed = _model()
    gparams = [p for p in graphed.parameters() if p.requires_grad]
    for p, s in zip(gparams, start):
        assert torch.equal(p, s)              # same seed, same adapters
    step = GraphedTrainStep(graphed, make(gparams), batches[0], warmup=0)
    got = [float(step(ids)) for ids in batches]
    if kind == "adamw":
        assert eager_loss[6] < eager_loss[0] and eager_loss[7] < eager_loss[1]     # it trains
    torch.testing.assert_close(torch.tensor(got), torch.tensor(eager_loss), rtol=2e-2, atol=2e-2)
    if kind == "sgd":
        # the summed update of every adapter matrix (Adam's sign-like steps make single elements with noise-sized
        # gradients incomparable; SGD's are proportional to the gradient)
        moved = 0
        for a, b, s in zip(gparams, params, start):
            da, db = (a - s).float(), (b - s).float()
            if db.norm() > 0:
                moved += 1
                assert (da - db).norm() <= 0.1 * db.norm() + 1e-6
        assert moved >= len(params) // 2
    with pytest.raises(ValueError):
        step(batches[0][:, :40])


def test_graphed_step_with_accumulation_and_padding():
    """two micro-batches per optimizer step, padded batches with labels and an attention mask, gradient clipping:
    the graph ends after backward, the rest runs eagerly; same losses and updates as the eager loop"""
    from bigdl_amd.transformers.graph_train import GraphedTrainStep
    gen = torch.Generator().manual_seed(4)
    data = []
    for i in range(6):
        ids = torch.randint(3, 512, (2, 40), generator=gen)
        mask = torch.ones_like(ids)
        mask[1, 40 - 5 - i:] = 0                       # right padding of different lengths
        labels = ids.masked_fill(mask == 0, -100)
        data.append((ids.cuda(), labels.cuda(), mask.cuda()))
    eager = _model()
    params = [p for p in eager.parameters() if p.requires_grad]
    start = [p.detach().clone() for p in params]
    opt = torch.optim.SGD(params, lr=0.05)
    eager_loss = []
    for i, (ids, labels, mask) in enumerate(data):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = eager(input_ids=ids, labels=labels, attention_mask=mask).loss
        (loss / 2).backward()
        eager_loss.append(float(loss.detach()))
        if i % 2 == 1:
            torch.nn.utils.clip_grad_norm_(params, 0.5)
            opt.step()
            opt.zero_grad(set_to_none=True)
    graphed = _model()
    gparams = [p for p in graphed.parameters() if p.requires_grad]
    step = GraphedTrainStep(graphed, torch.optim.SGD(gparams, lr=0.05), data[0][0], example_labels=data[0][1],
                            example_mask=data[0][2], warmup=0, grad_accum=2, max_grad_norm=0.5)
    got, stepped = [], []
    for ids, labels, mask in data:
        got.append(float(step(ids, labels, mask)))
        stepped.append(step.stepped)
    assert stepped == [False, True] * 3
    torch.testing.assert_close(torch.tensor(got), torch.tensor(eager_loss), rtol=2e-2, atol=2e-2)
    for a, b, s in zip(gparams, params, start):
        da, db = (a - s).float(), (b - s).float()
        if db.norm() > 0:
            assert (da - db).norm() <= 0.1 * db.norm() + 1e-6
    with pytest.raises(ValueError):
        step(data[0][0])                                # labels / mask are part of the captured step


def test_graphed_step_needs_capturable_optimizer():
    from bigdl_amd.transformers.graph_train import GraphedTrainStep
    model = _model()
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    with pytest.raises(ValueError):
        GraphedTrainStep(model, opt, torch.randint(3, 512, (2, 32)).cuda())


def test_graphed_step_refuses_activation_checkpointing():
    from bigdl_amd.transformers.graph_train import GraphedTrainStep
    model = _model()
    inner = model
    for _ in range(4):                       # the wrapper(s) around the HuggingFace model
        if hasattr(inner, "gradient_checkpointing_enable"):
            break
        inner = getattr(inner, "base_model", getattr(inner, "model", inner))
    inner.gradient_checkpointing_enable()
    with pytest.raises(ValueError, match="checkpointing"):
        GraphedTrainStep(model, torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.1),
                         torch.randint(3, 512, (2, 32)).cuda())
