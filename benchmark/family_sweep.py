"""Every causal-LM (or seq2seq) class of the installed ``transformers`` through ``optimize_model``: a tiny random model per
``model_type`` is built from a generic set of config arguments, converted, and its logits compared with the fp32 model's
(and, on the GPU, a short ``generate`` is run).  One JSON line per family: ``rel_err`` or the stage that failed
(``build`` / ``ref``: the generic config does not fit that family, nothing was tested; ``convert`` / ``forward`` /
``generate``: a defect here).  This is how the guards in ``transformers/convert.py`` (weight-dtype casts) and
``models/hf_patch.py`` (norm / MLP rebinding checks, attention registry) were found.

    python benchmark/family_sweep.py [--device cuda] [--low-bit sym_int8] [--seq2seq] [--only llama,mamba] [--max-params 3e7]
"""
import argparse
import copy
import inspect
import json
import logging
import signal
import sys
import os
import warnings

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

COMMON = dict(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
              num_key_value_heads=2, max_position_embeddings=128, n_embd=128, n_layer=2, n_head=4, n_positions=128,
              d_model=128, ffn_dim=256, n_inner=256, head_dim=32, num_experts=4, num_local_experts=4, num_experts_per_tok=2,
              moe_intermediate_size=64, pad_token_id=0, bos_token_id=1, eos_token_id=2)


class _Timeout(Exception):
    pass


def _alarm(signum, frame):
    raise _Timeout()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--low-bit", default="sym_int8")
    ap.add_argument("--seq2seq", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--max-params", type=float, default=3e7)
    ap.add_argument("--tol", type=float, default=0.08)
    ap.add_argument("--plain", action="store_true", help="no conversion: the HF model itself in fp16 on the device (what "
                                                         "half precision alone does to a family, e.g. MoE routing flips)")
    a = ap.parse_args()
    warnings.filterwarnings("ignore")
    logging.disable(logging.WARNING)
    from transformers import AutoModelForCausalLM, AutoModelForSeq2SeqLM
    from transformers.models.auto.configuration_auto import CONFIG_MAPPING
    from transformers.models.auto.modeling_auto import (MODEL_FOR_CAUSAL_LM_MAPPING_NAMES,
                                                        MODEL_FOR_SEQ_TO_SEQ_CAUSAL_LM_MAPPING_NAMES)
    from bigdl_amd.optimize import optimize_model
    auto = AutoModelForSeq2SeqLM if a.seq2seq else AutoModelForCausalLM
    names = sorted(MODEL_FOR_SEQ_TO_SEQ_CAUSAL_LM_MAPPING_NAMES if a.seq2seq else MODEL_FOR_CAUSAL_LM_MAPPING_NAMES)
    only = set(filter(None, a.only.split(",")))
    signal.signal(signal.SIGALRM, _alarm)
    bad = 0
    for mt in names:
        if only and mt not in only:
            continue
        rec, stage = {"model_type": mt}, "build"
        signal.alarm(90)
        try:
            torch.manual_seed(0)
            cfg_cls = CONFIG_MAPPING[mt]
            cfg = cfg_cls(**{k: v for k, v in COMMON.items() if k in inspect.signature(cfg_cls.__init__).parameters})
            with torch.device("meta"):
                n = sum(p.numel() for p in auto.from_config(cfg).parameters())
            if n > a.max_params:
                rec["skipped"] = "too big with default sizes"
                continue
            m = auto.from_config(cfg).eval()
            ids = torch.randint(3, 320, (2, 10))
            call = (lambda mod, i: mod(input_ids=i, decoder_input_ids=i[:, :5]).logits) if a.seq2seq else (lambda mod, i: mod(i).logits)
            with torch.no_grad():
                stage = "ref"
                ref = call(m, ids)
                stage = "convert"
                q = copy.deepcopy(m) if a.plain else optimize_model(copy.deepcopy(m), low_bit=a.low_bit)
                if a.device != "cpu":
                    q = q.half().to(a.device)
                stage = "forward"
                out = call(q, ids.to(a.device)).float().cpu()
                if out.shape != ref.shape:
                    rec["failed"], rec["error"] = "forward", f"logits {tuple(out.shape)} instead of {tuple(ref.shape)}"
                    bad += 1
                    continue
                rec["rel_err"] = round(((out - ref).abs().max() / ref.abs().max().clamp_min(1e-9)).item(), 5)
                if not rec["rel_err"] < a.tol:          # (nan fails too)
                    rec["off"] = True
                    bad += 1
                if a.device != "cpu" and not a.seq2seq:
                    stage = "generate"
                    g = q.generate(ids.to(a.device), max_new_tokens=4, do_sample=False)
                    rec["generated"] = list(g.shape)
        except _Timeout:
            rec["failed"], rec["error"] = stage, "timeout"
            bad += stage in ("convert", "forward", "generate")
        except Exception as e:          # noqa: BLE001
            rec["failed"], rec["error"] = stage, f"{type(e).__name__}: {str(e)[:160]}"
            bad += stage in ("convert", "forward", "generate")
        finally:
            signal.alarm(0)
            print(json.dumps(rec), flush=True)
    print(json.dumps({"defects": bad}), flush=True)


if __name__ == "__main__":
    main()
