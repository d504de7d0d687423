"""torch.library fake ops, perplexity harness, benchmark config."""
import os
import sys

import torch
import yaml
from transformers import LlamaConfig, LlamaForCausalLM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registered_ops_trace_on_meta():
    from bigdl_amd.ops import torch_ops
    names = torch_ops.registered_ops()
    for n in ("forward_new", "dequant", "mlp_forward", "dequantize_rows", "rms_norm", "layer_norm", "rotary_half_inplaced",
              "quantize_key_value", "moe_softmax_topk", "repetition_penalty_logits_process_inplaced"):
        assert n in names
    x = torch.empty(2, 5, 256, device="meta", dtype=torch.float16)
    w = torch.empty(4096, device="meta", dtype=torch.uint8)
    assert torch.ops.bigdl_amd.forward_new(x, w, 2, 64).shape == (2, 5, 64)
    assert torch.ops.bigdl_amd.mlp_forward(x[0, :1], w, w, 1, 256, 512, 1, 2).shape == (1, 512)
    ids, wt = torch.ops.bigdl_amd.moe_softmax_topk(torch.empty(7, 8, device="meta"), 2, True)
    assert ids.shape == (7, 2) and ids.dtype == torch.int32 and wt.dtype == torch.float32


def test_perplexity_low_bit_close_to_dense():
    sys.path.insert(0, os.path.join(ROOT, "benchmark", "perplexity"))
    from ppl import BigDLPPL
    from bigdl_amd.optimize import optimize_model
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=256, tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg).eval()
    toks = torch.randint(0, 320, (300,))
    import copy
    dense = BigDLPPL(model=copy.deepcopy(m)).perplexity_hf(toks, seq_len=64)
    q8 = BigDLPPL(model=optimize_model(copy.deepcopy(m), low_bit="sym_int8")).perplexity_hf(toks, seq_len=64)
    q4 = BigDLPPL(model=optimize_model(copy.deepcopy(m), low_bit="sym_int4")).perplexity_hf(toks, seq_len=64)
    assert 200 < dense < 500                       # a random model is near uniform over 320 tokens
    assert abs(q8 - dense) / dense < 0.01 and abs(q4 - dense) / dense < 0.05


def test_benchmark_config_schema():
    conf = yaml.safe_load(open(os.path.join(ROOT, "benchmark", "all_in_one", "config.yaml")))
    for k in ("repo_id", "local_model_hub", "warm_up", "num_trials", "num_beams", "low_bit", "batch_size", "in_out_pairs",
              "test_api", "cpu_embedding", "streaming"):
        assert k in conf
    assert "1024-128" in conf["in_out_pairs"]


def test_all_in_one_harness_on_a_synthetic_model(tmp_path):
    """benchmark/all_in_one/run.py with a ``synthetic:`` repo id through a HuggingFace front-door mode on the host: the
    reference's CSV columns, one row per in / out pair (no checkpoint needed)"""
    import csv
    import subprocess
    import sys
    cfg = tmp_path / "config.yaml"
    cfg.write_text(yaml.safe_dump({"repo_id": ["synthetic:tiny"], "local_model_hub": "", "warm_up": 1, "num_trials": 1,
                                   "num_beams": 1, "low_bit": "sym_int4", "batch_size": 1, "in_out_pairs": ["8-4", "16-4"],
                                   "test_api": ["transformer_int4"], "cpu_embedding": False, "streaming": False,
                                   "use_fp16_torch_dtype": True, "task": "continuation"}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "benchmark", "all_in_one", "run.py"), "-c", str(cfg), "-o",
                        str(tmp_path)], capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    files = [f for f in os.listdir(tmp_path) if f.endswith(".csv")]
    assert len(files) == 1 and files[0].startswith("transformer_int4-results-")
    rows = list(csv.DictReader(open(tmp_path / files[0])))
    assert [x["input/output tokens"] for x in rows] == ["8-4", "16-4"]
    assert all(float(x["2+ avg latency (ms/token)"]) > 0 and x["actual input/output tokens"] == x["input/output tokens"]
               for x in rows)


def test_llm_patch_rebinds_names_for_unmodified_scripts():
    """After llm_patch(train=True) a stock script's ``from transformers import AutoModelForCausalLM`` and
    ``from peft import LoraConfig, get_peft_model`` resolve to this library (PEFT itself is optional)."""
    import subprocess
    import sys
    code = (
        "from ipex_llm import llm_patch\n"
        "llm_patch(train=True)\n"
        "import transformers\n"
        "from transformers import AutoModelForCausalLM, TrainingArguments\n"
        "from peft import LoraConfig, get_peft_model, prepare_model_for_kbit_training\n"
        "assert AutoModelForCausalLM.__module__.startswith('bigdl_amd'), AutoModelForCausalLM\n"
        "assert TrainingArguments.__module__.startswith('bigdl_amd'), TrainingArguments\n"
        "assert get_peft_model.__module__.startswith('bigdl_amd')\n"
        "from ipex_llm.llm_patching import llm_unpatch\n"
        "llm_unpatch()\n"
        "assert transformers.AutoModelForCausalLM.__module__.startswith('transformers')\n"
        "print('ok')\n")
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
