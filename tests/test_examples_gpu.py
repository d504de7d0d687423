"""Example scripts end to end on the GPU against a tiny random checkpoint (the reference runs its examples the same
way, python/llm/test/run-llm-example-tests-gpu.sh).  Each script runs in its own interpreter, exit status 0 and a
non-empty output are required."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import Whitespace
    from transformers import LlamaConfig, LlamaForCausalLM, PreTrainedTokenizerFast
    d = tmp_path_factory.mktemp("tiny_llama")
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=2048, tie_word_embeddings=False)
    model = LlamaForCausalLM(cfg).eval()
    model.generation_config.eos_token_id = None
    model.save_pretrained(str(d), safe_serialization=True)
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    vocab.update({f"w{i}": i + 3 for i in range(317)})
    t = Tokenizer(WordLevel(vocab, unk_token="<unk>"))
    t.pre_tokenizer = Whitespace()
    tok = PreTrainedTokenizerFast(tokenizer_object=t, bos_token="<s>", eos_token="</s>", unk_token="<unk>", pad_token="</s>")
    tok.save_pretrained(str(d))
    prompt = d / "prompt.txt"
    prompt.write_text(" ".join(f"w{(7 * i) % 300}" for i in range(400)))
    alpaca = d / "alpaca.json"
    alpaca.write_text(json.dumps([{"instruction": f"w{i} w{i + 1}", "input": "", "output": f"w{i + 2} w{i + 3}"}
                                  for i in range(16)]))
    (d / "pairs.jsonl").write_text("\n".join(json.dumps({"prompt": f"w{i} w{i + 1}", "chosen": f"w{i + 2} w{i + 3}",
                                                          "rejected": f"w{i + 9} w{i + 5}"}) for i in range(8)))
    (d / "lines.txt").write_text("\n".join(" ".join(f"w{(3 * i + j) % 300}" for j in range(12)) for i in range(32)))
    return d


def _torchrun(script, *args, timeout=300):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1",
                        "--nproc-per-node", "1", os.path.join(ROOT, "examples", "GPU", script), *map(str, args)], env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{script} failed:\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}"
    return r.stdout


def _run(script, *args, timeout=240):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "GPU", script), *map(str, args)], env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{script} failed:\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}"
    assert r.stdout.strip(), f"{script} printed nothing"
    return r.stdout


def test_generate_and_data_types(ckpt):
    _run("HuggingFace/generate.py", "--repo-id-or-model-path", ckpt, "--prompt", "w1 w2 w3", "--n-predict", 8)
    out = _run("HuggingFace/More-Data-Types/generate.py", "--repo-id-or-model-path", ckpt, "--low-bit", "nf4", "fp8_e5m2",
               "--prompt", "w1 w2 w3", "--n-predict", 8)
    assert "nf4" in out and "fp8_e5m2" in out


def test_save_load_and_generic_optimize(ckpt, tmp_path):
    _run("Save-Load/save_load.py", "--repo-id-or-model-path", ckpt, "--save-path", tmp_path / "lb")
    assert (tmp_path / "lb" / "load_keys.json").exists()
    _run("PyTorch-Models/optimize_any_model.py", "--repo-id-or-model-path", ckpt, "--save-dir", tmp_path / "opt")


def test_lookup_speculative_long_context(ckpt):
    _run("Lookahead/lookup.py", "--repo-id-or-model-path", ckpt, "--prompt-file", ckpt / "prompt.txt", "--n-predict", 16)
    _run("Speculative-Decoding/speculative.py", "--repo-id-or-model-path", ckpt, "--prompt", "w1 w2 w3", "--n-predict", 16)
    _run("Long-Context/generate.py", "--repo-id-or-model-path", ckpt, "--prompt-file", ckpt / "prompt.txt", "--n-predict", 8,
         "--fp8-kv")


def test_qlora_finetuning_example(ckpt, tmp_path):
    _run("LLM-Finetuning/QLoRA/alpaca_qlora_finetuning.py", "--base_model", ckpt, "--data_path", ckpt / "alpaca.json",
         "--output_dir", tmp_path / "out", "--micro_batch_size", 2, "--batch_size", 4, "--cutoff_len", 32,
         "--lora_target_modules", "q_proj", "v_proj")


def test_graph_mode_example():
    out = _run("GraphMode/graph_decode.py", "--n-predict", 16)
    assert "ms" in out


def test_more_finetuning_examples(ckpt, tmp_path):
    _run("LLM-Finetuning/DPO/dpo_finetuning.py", "--repo-id-or-model-path", ckpt, "--dataset", ckpt / "pairs.jsonl",
         "--output-path", tmp_path / "dpo", "--micro-batch", 2, "--grad-accum", 1, "--max-steps", 2)
    _run("LLM-Finetuning/GaLore/galore_finetuning.py", "--repo-id-or-model-path", ckpt, "--data-path", ckpt / "alpaca.json",
         "--output-dir", tmp_path / "galore", "--rank", 8, "--max-steps", 3, "--cutoff-len", 32)
    _run("LLM-Finetuning/LISA/lisa_finetuning.py", "--base_model", ckpt, "--text_file", ckpt / "lines.txt", "--steps", 4,
         "--lisa_interval_steps", 2)
    _torchrun("LLM-Finetuning/ZeRO/qlora_zero.py", "--base", ckpt, "--stage", 3, "--steps", 2, "--micro-batch", 2,
              "--seq-len", 16)


def test_parallel_examples_with_one_rank(ckpt):
    """world size 1: the sharding / staging code paths run, the collectives are no-ops"""
    out = _torchrun("Tensor-Parallel/generate.py", "--repo-id-or-model-path", ckpt, "--prompt", "w1 w2 w3", "--n-predict", 8)
    assert out.strip()
    out = _torchrun("Pipeline-Parallel-Inference/generate.py", "--repo-id-or-model-path", ckpt, "--prompt", "w1 w2 w3",
                    "--n-predict", 8, "--gpu-num", 1)
    assert out.strip()


def test_unmodified_hf_peft_script(ckpt, tmp_path):
    """llm_patch(train=True) + a stock transformers / peft training script (200 optimizer steps on the tiny model)"""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "GPU", "LLM-Finetuning", "HF-PEFT",
                                                     "alpaca_lora_unmodified.py"), str(ckpt), str(ckpt / "lines.txt")],
                       env=env, capture_output=True, text=True, timeout=500, cwd=str(tmp_path))
    assert r.returncode == 0, f"{r.stdout[-1500:]}\n{r.stderr[-3000:]}"


@pytest.fixture(scope="module")
def moe_ckpt(ckpt, tmp_path_factory):
    import shutil
    from transformers import MixtralConfig, MixtralForCausalLM
    d = tmp_path_factory.mktemp("tiny_mixtral")
    torch.manual_seed(0)
    cfg = MixtralConfig(vocab_size=320, hidden_size=256, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                        num_key_value_heads=2, num_local_experts=4, num_experts_per_tok=2, max_position_embeddings=512)
    model = MixtralForCausalLM(cfg).eval()
    model.generation_config.eos_token_id = None
    model.save_pretrained(str(d), safe_serialization=True)
    for f in os.listdir(ckpt):
        if "token" in f:
            shutil.copy(os.path.join(ckpt, f), os.path.join(d, f))
    return d


def test_moe_examples(moe_ckpt):
    _run("CPU-GPU-Hybrid/generate.py", "--repo-id-or-model-path", moe_ckpt, "--prompt", "w1 w2 w3", "--n-predict", 6,
         "--gpu-layers", 1)
    for mode in ("a2a", "allreduce"):
        out = _torchrun("MoE/mixtral_expert_parallel.py", "--repo-id-or-model-path", moe_ckpt, "--mode", mode,
                        "--n-predict", 8)
        assert out.strip()


def _serve_and_query(script, *args, port):
    """start a server example, wait for /health, stream one request, stop the exact process that was started"""
    import json as js
    import time
    import urllib.request
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "examples", "GPU", script), *map(str, args), "--port", str(port)],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        deadline = time.time() + 120
        while True:
            assert p.poll() is None, "server exited:\n" + p.stdout.read()[-3000:]
            try:
                urllib.request.urlopen(f"http://127.0.0.1:{port}/health", timeout=2).read()
                break
            except Exception:
                assert time.time() < deadline, "server did not come up"
                time.sleep(1.0)
        body = js.dumps({"inputs": "w1 w2 w3 w4", "parameters": {"max_new_tokens": 8}}).encode()
        req = urllib.request.Request(f"http://127.0.0.1:{port}/generate", data=body, headers={"Content-Type": "application/json"})
        out = js.loads(urllib.request.urlopen(req, timeout=60).read())
        assert out["generate_tokens"] >= 1
        body = js.dumps({"model": "m", "prompt": "w5 w6", "max_tokens": 6}).encode()
        req = urllib.request.Request(f"http://127.0.0.1:{port}/v1/completions", data=body, headers={"Content-Type": "application/json"})
        out = js.loads(urllib.request.urlopen(req, timeout=60).read())
        assert out["choices"][0]["text"] is not None
    finally:
        p.terminate()
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()


def test_serving_examples(ckpt):
    _serve_and_query("Lightweight-Serving/lightweight_serving.py", "--repo-id-or-model-path", ckpt, port=18431)
    _serve_and_query("Continuous-Batching/serve.py", "--repo-id-or-model-path", ckpt, "--max-num-seqs", 4, "--max-seq-len", 256,
                     port=18432)


def test_gguf_example(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("gguf_helpers", os.path.join(ROOT, "tests", "test_gguf_cpu.py"))
    H = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(H)
    path = str(tmp_path / "tiny.gguf")
    H._write_llama(H._tiny_llama(), path, H.G.GGML_Q4_0)
    out = _run("GGUF/generate.py", "--model", path, "--prompt", "hello", "--n-predict", 6)
    assert out.strip()


def test_all_in_one_benchmark_harness(ckpt, tmp_path):
    """benchmark/all_in_one/run.py (the reference's metric definition) over the HF-model test_api modes"""
    import csv
    import yaml
    apis = ["transformer_int4_gpu", "transformer_int4_fp16_gpu", "optimize_model_gpu", "bigdl_fp16_gpu", "lookup_gpu",
            "speculative_gpu", "ipex_fp16_gpu", "transformer_int4_fp16_gpu_win"]     # (_win: a reference config name)
    conf = dict(repo_id=[os.path.basename(str(ckpt))], local_model_hub=os.path.dirname(str(ckpt)), warm_up=1, num_trials=2,
                num_beams=1, low_bit="sym_int4", batch_size=1, in_out_pairs=["32-8"], test_api=apis, cpu_embedding=False,
                streaming=False, use_fp16_torch_dtype=True, quantize_kv_cache=False, task="continuation")
    cfg = tmp_path / "config.yaml"
    cfg.write_text(yaml.safe_dump(conf))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "benchmark", "all_in_one", "run.py"), "-c", str(cfg), "-o",
                        str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"{r.stdout[-2000:]}\n{r.stderr[-3000:]}"
    files = [f for f in os.listdir(tmp_path) if f.endswith(".csv")]
    assert len(files) == len(apis), files
    for f in files:
        rows = list(csv.DictReader(open(tmp_path / f)))
        assert len(rows) == 1 and float(rows[0]["1st token avg latency (ms)"]) > 0 and float(rows[0]["2+ avg latency (ms/token)"]) > 0


def test_all_in_one_tensor_parallel_mode(ckpt, tmp_path):
    """the reference's deepspeed_optimize_model_gpu mode: here this library's own sharder under torchrun"""
    import yaml
    conf = dict(repo_id=[os.path.basename(str(ckpt))], local_model_hub=os.path.dirname(str(ckpt)), warm_up=1, num_trials=1,
                num_beams=1, low_bit="sym_int4", batch_size=1, in_out_pairs=["32-8"], test_api=["deepspeed_optimize_model_gpu"],
                cpu_embedding=False, streaming=False, use_fp16_torch_dtype=True, quantize_kv_cache=False, task="continuation")
    cfg = tmp_path / "config.yaml"
    cfg.write_text(yaml.safe_dump(conf))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1",
                        "--nproc-per-node", "1", os.path.join(ROOT, "benchmark", "all_in_one", "run.py"), "-c", str(cfg), "-o",
                        str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"{r.stdout[-2000:]}\n{r.stderr[-3000:]}"
    assert any(f.startswith("deepspeed_optimize_model_gpu-results") for f in os.listdir(tmp_path))


def test_engine_speculative_example_with_layer_skipping_draft(ckpt):
    out = _run("Speculative-Decoding/engine_speculative.py", "--repo-id-or-model-path", ckpt, "--prompt", "w1 w2 w3 w4",
               "--n-predict", 16, "--max-draft", 3, "--draft-layers", 1)
    assert "drafted" in out and "tokens per round" in out


def test_eagle_example_with_stand_in_package(ckpt, tmp_path):
    """Speculative-Decoding/EAGLE against a stand-in ``eagle`` package (the real one is not installable here: parity
    with it is unpinned): ``EaModel`` with a base model, a one-layer draft head and ``eagenerate``; the script swaps in
    the low-bit base model, converts the draft head and writes the answer file that speed.py summarizes."""
    pkg = tmp_path / "site" / "eagle" / "model"
    pkg.mkdir(parents=True)
    (tmp_path / "site" / "eagle" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "ea_model.py").write_text('''
import torch
from torch import nn
from transformers import AutoModelForCausalLM


class EaLayer(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.fc = nn.Linear(2 * hidden, hidden, bias=False)
        self.tree_mask_init = torch.eye(4)
        self.position_ids = torch.zeros(4, dtype=torch.long)


class EaModel(nn.Module):
    def __init__(self, base):
        super().__init__()
        self.base_model = base
        self.ea_layer = EaLayer(base.config.hidden_size)

    @classmethod
    def from_pretrained(cls, base_model_path=None, ea_model_path=None, total_token=60, depth=5, top_k=10, **kw):
        kw.pop("low_cpu_mem_usage", None)
        return cls(AutoModelForCausalLM.from_pretrained(base_model_path, **kw))

    def eagenerate(self, input_ids, temperature=0.0, max_new_tokens=32, log=False):
        assert type(self.ea_layer.fc).__name__ == "LowBitLinear" and self.ea_layer.tree_mask_init.is_cuda
        out = self.base_model.generate(input_ids, max_new_tokens=max_new_tokens, do_sample=False)
        return (out, out.shape[1] - input_ids.shape[1], 1) if log else out
''')
    q = tmp_path / "q.jsonl"
    q.write_text("\n".join(json.dumps({"question_id": i, "turns": [f"w{i} w{i + 1} w{i + 2}", f"w{i + 5} w{i + 6}"]})
                           for i in range(3)))
    ans = tmp_path / "answers.jsonl"
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, str(tmp_path / "site"), os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "GPU", "Speculative-Decoding", "EAGLE", "eagle_generate.py"),
                        "--base-model-path", str(ckpt), "--ea-model-path", str(ckpt), "--question-file", str(q),
                        "--answer-file", str(ans), "--max-new-token", "8"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    rows = [json.loads(line) for line in ans.read_text().splitlines()]
    assert len(rows) == 2 and all(sum(x["choices"][0]["new_tokens"]) == 16 for x in rows)      # the first question warms up
    out = _run("Speculative-Decoding/EAGLE/speed.py", "--jsonl-file", ans)
    assert json.loads(out)["total_tokens"] == 32
