"""Every low-bit format through the checkpoint cycle on the device: tiny llama + tiny mixtral -> optimize_model(fmt) ->
save_low_bit -> load_low_bit -> .half().to(device) -> logits equal to those of the model that was saved (moved to the
device the same way), and GPU -> CPU -> save again gives the same bytes.  One JSON line per (family, format).

    python benchmark/format_roundtrip_sweep.py [--device cuda]
"""
import argparse
import copy
import json
import logging
import os
import sys
import tempfile
import warnings

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

FORMATS = ["sym_int4", "asym_int4", "sym_int5", "asym_int5", "sym_int8", "nf4", "nf3", "fp4", "fp6", "fp8_e4m3", "fp8_e5m2",
           "q4_k", "q5_k", "q6_k", "q2_k", "fp6_k", "woq_int4", "gguf_iq2_xxs", "gguf_iq2_xs", "gguf_iq1_s", "mixed_fp4",
           "mixed_fp8", "fp16", "bf16"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cpu")
    a = ap.parse_args()
    warnings.filterwarnings("ignore")
    logging.disable(logging.WARNING)
    from transformers import LlamaConfig, LlamaForCausalLM, MixtralConfig, MixtralForCausalLM
    from bigdl_amd.optimize import load_low_bit, low_memory_init, optimize_model
    kw = dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
              num_key_value_heads=2, max_position_embeddings=128)
    fams = {"llama": (LlamaConfig(**kw), LlamaForCausalLM),
            "mixtral": (MixtralConfig(num_local_experts=4, num_experts_per_tok=2, **kw), MixtralForCausalLM)}
    bad = 0
    for fam, (cfg, cls) in fams.items():
        torch.manual_seed(0)
        base = cls(cfg).eval()
        ids = torch.randint(3, 320, (2, 10))
        for fmt in FORMATS:
            rec = {"family": fam, "format": fmt}
            try:
                with torch.no_grad():
                    q = optimize_model(copy.deepcopy(base), low_bit=fmt)
                    d = tempfile.mkdtemp(prefix="rt_")
                    q.save_low_bit(d)
                    with low_memory_init():
                        shell = cls(cfg)
                    r = load_low_bit(shell, d)
                    if a.device != "cpu":
                        q, r = q.half().to(a.device), r.half().to(a.device)
                    lq, lr = q(ids.to(a.device)).logits.float(), r(ids.to(a.device)).logits.float()
                    rec["max_abs_diff"] = round((lq - lr).abs().max().item(), 6)
                    rec["finite"] = bool(torch.isfinite(lr).all())
                    if a.device != "cpu":
                        d2 = tempfile.mkdtemp(prefix="rt2_")
                        r.save_low_bit(d2)                      # device tiles -> host blocks -> file
                        s1 = torch.load(os.path.join(d, "pytorch_model.bin"), weights_only=True) if os.path.exists(
                            os.path.join(d, "pytorch_model.bin")) else None
                        if s1 is not None:
                            s2 = torch.load(os.path.join(d2, "pytorch_model.bin"), weights_only=True)
                            rec["bytes_equal"] = all(torch.equal(s1[k].cpu(), s2[k].cpu()) for k in s1
                                                     if s1[k].dtype == torch.uint8)
                tol = 2e-2 if fmt in ("fp16", "bf16") else 1e-3       # (16-bit weights: the shell's dtype handling differs by an ulp)
                if rec["max_abs_diff"] > tol or not rec["finite"] or rec.get("bytes_equal") is False:
                    rec["off"] = True
                    bad += 1
            except Exception as e:          # noqa: BLE001
                rec["failed"] = f"{type(e).__name__}: {str(e)[:200]}"
                bad += 1
            print(json.dumps(rec), flush=True)
    print(json.dumps({"defects": bad}), flush=True)


if __name__ == "__main__":
    main()
