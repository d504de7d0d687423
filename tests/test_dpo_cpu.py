"""DPO on a low-bit base with LoRA adapters: the adapters-off forward is the reference policy."""
import torch
from transformers import LlamaConfig, LlamaForCausalLM

from bigdl_amd import optimize_model
from bigdl_amd.transformers.dpo import DPOConfig, DPOTrainer, dpo_loss, sequence_logps
from bigdl_amd.transformers.qlora import LoraConfig, get_peft_model


def test_sequence_logps_and_losses():
    torch.manual_seed(0)
    logits = torch.randn(2, 6, 11)
    labels = torch.tensor([[-100, -100, 3, 4, 5, -100], [-100, 1, 2, -100, -100, -100]])
    got = sequence_logps(logits, labels)
    lp = logits.log_softmax(-1)
    want0 = lp[0, 1, 3] + lp[0, 2, 4] + lp[0, 3, 5]
    want1 = lp[1, 0, 1] + lp[1, 1, 2]
    torch.testing.assert_close(got, torch.stack([want0, want1]))
    pc, pr, rc, rr = torch.tensor([-5.0]), torch.tensor([-9.0]), torch.tensor([-6.0]), torch.tensor([-8.0])
    loss, c, r = dpo_loss(pc, pr, rc, rr, beta=0.5)
    torch.testing.assert_close(loss, -torch.nn.functional.logsigmoid(torch.tensor([0.5 * 2.0])))
    assert c.item() == 0.5 and r.item() == -0.5
    assert dpo_loss(pc, pr, rc, rr, beta=0.5, loss_type="hinge")[0].item() == 0.0
    torch.testing.assert_close(dpo_loss(pc, pr, rc, rr, beta=0.5, loss_type="ipo")[0], torch.tensor([1.0]))


def test_dpo_training_raises_the_preference_margin():
    torch.manual_seed(1)
    cfg = LlamaConfig(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=4, max_position_embeddings=64)
    base = optimize_model(LlamaForCausalLM(cfg), low_bit="nf4", optimize_llm=False)
    model = get_peft_model(base, LoraConfig(r=4, lora_alpha=8, target_modules=["q_proj", "v_proj", "down_proj"],
                                            training_mode="qlora"))
    prompt = torch.randint(0, 64, (4, 5))
    chosen, rejected = torch.randint(0, 64, (4, 4)), torch.randint(0, 64, (4, 4))
    ids = torch.cat([torch.cat([prompt, chosen], 1), torch.cat([prompt, rejected], 1)], 0)
    labels = ids.clone()
    labels[:, :5] = -100
    batch = {"input_ids": ids, "labels": labels, "attention_mask": torch.ones_like(ids)}
    tr = DPOTrainer(model, DPOConfig(beta=0.5, learning_rate=5e-3, max_grad_norm=1.0))
    ref0 = tr.reference_logps(batch)
    first = tr.step(batch)
    assert abs(first["loss"] - 0.6931) < 1e-3 and first["rewards/margins"] == 0.0   # lora_B = 0: policy == reference
    for _ in range(15):
        last = tr.step(batch)
    assert last["loss"] < first["loss"] - 0.05 and last["rewards/margins"] > 0.1
    torch.testing.assert_close(tr.reference_logps(batch), ref0)      # the adapters-off reference policy has not moved
