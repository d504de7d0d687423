"""All-in-one latency benchmark: first-token / next-token latency and peak memory per (model, in-out pair, test_api).

Reference: python/llm/dev/benchmark/all-in-one/run.py (``run_model`` dispatch :145-205 over ~30 ``test_api`` modes, CSV
columns ``model, 1st token avg latency (ms), 2+ avg latency (ms/token), encoder time (ms), input/output tokens,
batch_size, actual input/output tokens, num_beams, low_bit, cpu_embedding, model loading time (s), peak mem (GB),
streaming, use_fp16_torch_dtype`` :2287-2359).  The timed quantities are defined identically: ``first_cost`` is the
wall time of the prefill step, ``rest_cost_mean`` the mean wall time of the following decode steps, each closed by a
device synchronize.
"""
from __future__ import annotations

import argparse
import csv
import gc
import os
import sys
import time
from datetime import date

import torch
import yaml

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))

COLUMNS = ["model", "1st token avg latency (ms)", "2+ avg latency (ms/token)", "encoder time (ms)", "input/output tokens",
           "batch_size", "actual input/output tokens", "num_beams", "low_bit", "cpu_embedding",
           "model loading time (s)", "peak mem (GB)", "streaming", "use_fp16_torch_dtype", "test_api"]

SYNTHETIC = {}


def _synthetic_cfg(name):
    from bigdl_amd.models import decoder as D
    table = {"llama2-7b": D.LLAMA2_7B, "tinyllama": D.TINY_LLAMA, "mixtral-8x7b": D.MIXTRAL_8X7B,
             "tiny": dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, num_kv_heads=2,
                          head_dim=64, rms_eps=1e-5, max_position=2048, model_type="llama")}      # test-size llama
    return D.DecoderConfig(**table[name])


def _prompt_ids(tokenizer, n, vocab, batch):
    g = torch.Generator().manual_seed(0)
    return torch.randint(10, min(vocab, 30000), (batch, n), generator=g)


def run_native_engine_gpu(repo_id, local_model_hub, in_out_pairs, warm_up, num_trials, low_bit, batch_size, conf):
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import build_random_decoder
    t0 = time.perf_counter()
    if repo_id.startswith("synthetic:"):
        cfg = _synthetic_cfg(repo_id.split(":", 1)[1])
        eng = build_random_decoder(cfg, Q.qtype_id(low_bit))
        vocab = cfg.vocab_size
    else:
        model, tok = _load_hf(repo_id, local_model_hub, low_bit, conf, "cuda")
        from bigdl_amd.models.hf_patch import get_native_engine
        eng = get_native_engine(model)
        vocab = model.config.vocab_size
    load = time.perf_counter() - t0
    rows = []
    for pair in in_out_pairs:
        i, o = (int(x) for x in pair.split("-"))
        ids = _prompt_ids(None, i, vocab, batch_size).cuda()
        firsts, rests, peaks = [], [], []
        for t in range(warm_up + num_trials):
            out = eng.generate(ids, max_new_tokens=o, kv_fp8=bool(conf.get("quantize_kv_cache", False)))
            if t >= warm_up:
                firsts.append(eng.first_cost)
                rests.append(eng.rest_cost_mean)
                peaks.append(eng.peak_memory)
        rows.append([repo_id, round(1e3 * sum(firsts) / len(firsts), 2), round(1e3 * sum(rests) / len(rests), 3), 0.0, pair,
                     batch_size, f"{i}-{out.shape[1] - i}", 1, low_bit, False, round(load, 2), round(max(peaks), 2), False,
                     True, "native_engine_gpu"])
    return rows


def _synthetic_hf(name, device):
    """random-init HuggingFace Llama of a named shape in fp16 (no checkpoints on the test boxes): what a checkpoint of
    that architecture gives ``from_pretrained``, drawn on the device and kept on the host"""
    from transformers import LlamaConfig, LlamaForCausalLM
    d = _synthetic_cfg(name)
    cfg = LlamaConfig(vocab_size=d.vocab_size, hidden_size=d.hidden_size, intermediate_size=d.intermediate_size,
                      num_hidden_layers=d.num_layers, num_attention_heads=d.num_heads, num_key_value_heads=d.num_kv_heads,
                      max_position_embeddings=d.max_position, rms_norm_eps=d.rms_eps, rope_theta=d.rope_theta,
                      tie_word_embeddings=False)
    with torch.device("meta"):
        model = LlamaForCausalLM(cfg)
    model = model.to_empty(device="cpu").half()
    gen_dev = device if device != "cpu" else "cpu"
    g = torch.Generator(device=gen_dev).manual_seed(0)
    with torch.no_grad():
        for _, p in model.named_parameters():
            if p.dim() == 1:
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, device=gen_dev, generator=g, dtype=torch.float32) * 0.02).half())
        for m in model.modules():          # buffers made on the meta device (rotary inv_freq) are uninitialised
            if hasattr(m, "inv_freq"):
                fresh = type(m)(cfg)
                m.inv_freq = fresh.inv_freq.float()
                if hasattr(m, "original_inv_freq"):
                    m.original_inv_freq = fresh.inv_freq.float()
    model.generation_config.eos_token_id = None
    model.generation_config.pad_token_id = 0
    return model.eval()


def _load_hf(repo_id, hub, low_bit, conf, device, plain=False, tensor_parallel=False, **extra):
    from transformers import AutoTokenizer
    from bigdl_amd.transformers import AutoModelForCausalLM
    if repo_id.startswith("synthetic:") and not tensor_parallel:
        model = _synthetic_hf(repo_id.split(":", 1)[1], device)
        if not plain:
            from bigdl_amd.optimize import optimize_model
            model = optimize_model(model, low_bit=low_bit, cpu_embedding=conf.get("cpu_embedding", False), **extra)
        if device != "cpu":
            model = model.half().to(device)
        return model.eval(), None
    path = os.path.join(hub, os.path.basename(repo_id)) if hub else repo_id
    dt = torch.float16 if conf.get("use_fp16_torch_dtype", True) and device != "cpu" else "auto"
    if plain:
        # unconverted HuggingFace model: the fp16 / bf16 baselines of the reference (ipex_fp16_gpu, pytorch_autocast_bf16)
        import transformers
        model = transformers.AutoModelForCausalLM.from_pretrained(path, torch_dtype=torch.float16 if device != "cpu" else torch.bfloat16)
        tok = AutoTokenizer.from_pretrained(path) if os.path.exists(os.path.join(path, "tokenizer_config.json")) else None
        return model.to(device).eval(), tok
    if tensor_parallel:
        # the reference's deepspeed_optimize_model_gpu / deepspeed_transformer_int4_cpu: one model sharded over the
        # ranks of the job (torchrun --nproc-per-node N), all-reduce after the row-parallel linears
        from bigdl_amd.parallel import init_distributed
        from bigdl_amd.parallel.tensor_parallel import tensor_parallel as shard
        _, _, dev = init_distributed(backend=None if device != "cpu" else "gloo")
        model = AutoModelForCausalLM.from_pretrained(path, load_in_low_bit=low_bit, torch_dtype=dt, use_cache=True)
        model = shard(model)
        model = model.half().to(dev) if device != "cpu" else model
        if hasattr(model, "_hf_generate"):
            model.generate = model._hf_generate          # the sharded modules run through the HF loop, not the engine
        tok = AutoTokenizer.from_pretrained(path) if os.path.exists(os.path.join(path, "tokenizer_config.json")) else None
        return model.eval(), tok
    if os.path.exists(os.path.join(path, "bigdl_config.json")) or _is_low_bit_dir(path):
        model = AutoModelForCausalLM.load_low_bit(path, torch_dtype=dt, **extra)
    else:
        model = AutoModelForCausalLM.from_pretrained(path, load_in_low_bit=low_bit, torch_dtype=dt, use_cache=True,
                                                     cpu_embedding=conf.get("cpu_embedding", False), **extra)
    if device != "cpu":
        model = model.half().to(device)
    try:
        tok = AutoTokenizer.from_pretrained(path)
    except Exception:
        tok = None
    return model.eval(), tok


def _is_low_bit_dir(path):
    import json
    try:
        return "bigdl_transformers_low_bit" in json.load(open(os.path.join(path, "config.json")))
    except Exception:
        return False


def _run_hf_generate(repo_id, hub, in_out_pairs, warm_up, num_trials, low_bit, batch_size, conf, device, api,
                     gen_kwargs=None, load_kwargs=None):
    from bigdl_amd.utils.benchmark_util import BenchmarkWrapper
    t0 = time.perf_counter()
    model, tok = _load_hf(repo_id, hub, low_bit, conf, device, **(load_kwargs or {}))
    load = time.perf_counter() - t0
    if conf.get("quantize_kv_cache"):
        os.environ["IPEX_LLM_QUANTIZE_KV_CACHE"] = "1"
    wrapped = BenchmarkWrapper(model, do_print=False)
    rows = []
    for pair in in_out_pairs:
        i, o = (int(x) for x in pair.split("-"))
        ids = _prompt_ids(tok, i, model.config.vocab_size, batch_size).to(next(model.parameters()).device)
        firsts, rests = [], []
        for t in range(warm_up + num_trials):
            out = wrapped.generate(ids, max_new_tokens=o, min_new_tokens=o, do_sample=False, **(gen_kwargs or {}))
            if t >= warm_up:
                firsts.append(wrapped.first_cost)
                rests.append(wrapped.rest_cost_mean)
        rows.append([repo_id, round(1e3 * sum(firsts) / len(firsts), 2), round(1e3 * sum(rests) / len(rests), 3), 0.0, pair,
                     batch_size, f"{i}-{out.shape[1] - i}", conf.get("num_beams", 1), low_bit, conf.get("cpu_embedding", False),
                     round(load, 2), round(wrapped.peak_memory, 2), conf.get("streaming", False),
                     conf.get("use_fp16_torch_dtype", True), api])
    del model
    gc.collect()
    return rows


# names the reference's config files use for the same measurement on other hosts (reference run.py:145-205): the
# Windows variants differ only in how the process is set up there, the IPEX CPU modes are the CPU path of this package
API_ALIASES = {"bigdl_ipex_bf16": ("transformer_autocast_bf16", None), "bigdl_ipex_int4": ("transformer_int4", "sym_int4"),
               "bigdl_ipex_int8": ("transformer_int4", "sym_int8")}
UNSUPPORTED_APIS = ("transformers_int4_npu_win", "transformers_int4_npu_pipeline_win", "transformers_int4_loadlowbit_npu_win",
                    "transformers_openvino")


def run_model(repo_id, test_api, in_out_pairs, local_model_hub, warm_up, num_trials, low_bit, batch_size, conf):
    if test_api in UNSUPPORTED_APIS:
        raise ValueError(f"test_api {test_api} measures another runtime (Intel NPU / OpenVINO); nothing to run here")
    if test_api.endswith("_win"):
        test_api = test_api[:-len("_win")]
    if test_api in API_ALIASES:
        test_api, lb = API_ALIASES[test_api]
        low_bit = lb or low_bit
    gpu = dict(device="cuda")
    if test_api == "native_engine_gpu":
        return run_native_engine_gpu(repo_id, local_model_hub, in_out_pairs, warm_up, num_trials, low_bit, batch_size, conf)
    table = {
        "transformer_int4_gpu": dict(**gpu),
        "transformer_int4_fp16_gpu": dict(**gpu),
        "transformer_int4_loadlowbit_gpu": dict(**gpu),
        "transformer_int4_fp16_loadlowbit_gpu": dict(**gpu),
        "optimize_model_gpu": dict(**gpu),
        "transformer_autocast_bf16": dict(device="cpu"),
        "transformer_int4": dict(device="cpu"),
        "transformer_int4_loadlowbit": dict(device="cpu"),
        "optimize_model": dict(device="cpu"),
        "bigdl_fp16_gpu": dict(**gpu),
        "lookup_gpu": dict(**gpu, gen_kwargs=dict(lookahead=3, max_matching_ngram_size=2)),
        "speculative_gpu": dict(**gpu, load_kwargs=dict(speculative=True)),
        "speculative_cpu": dict(device="cpu", load_kwargs=dict(speculative=True)),
        "pipeline_parallel_gpu": dict(**gpu, load_kwargs=dict(pipeline_parallel_stages=int(os.environ.get("WORLD_SIZE", 1)))),
        "ipex_fp16_gpu": dict(**gpu, load_kwargs=dict(plain=True)),
        "pytorch_autocast_bf16": dict(device="cpu", load_kwargs=dict(plain=True)),
        "deepspeed_optimize_model_gpu": dict(**gpu, load_kwargs=dict(tensor_parallel=True)),
        "deepspeed_transformer_int4_cpu": dict(device="cpu", load_kwargs=dict(tensor_parallel=True)),
    }
    if test_api in table:
        kw = table[test_api]
        lb = "fp16" if test_api == "bigdl_fp16_gpu" else low_bit
        if test_api == "speculative_gpu":
            lb = "fp16"
        if test_api == "pipeline_parallel_gpu":
            from bigdl_amd.transformers import init_pipeline_parallel
            init_pipeline_parallel()
        return _run_hf_generate(repo_id, local_model_hub, in_out_pairs, warm_up, num_trials, lb, batch_size, conf,
                                kw["device"], test_api, kw.get("gen_kwargs"), kw.get("load_kwargs"))
    if test_api == "native_int4":
        from bigdl_amd.ggml.model import Llama
        path = os.path.join(local_model_hub, os.path.basename(repo_id)) if local_model_hub else repo_id
        t0 = time.perf_counter()
        llm = Llama(path)
        load = time.perf_counter() - t0
        rows = []
        for pair in in_out_pairs:
            i, o = (int(x) for x in pair.split("-"))
            toks = _prompt_ids(None, i, llm.n_vocab(), 1)[0].tolist()
            t1 = time.perf_counter()
            n = len(list(llm.generate(toks, temp=0.0, max_tokens=o)))
            dt = time.perf_counter() - t1
            rows.append([repo_id, round(1e3 * getattr(llm.model, "first_cost", 0.0), 2), round(1e3 * dt / max(1, n), 3), 0.0, pair,
                         1, f"{i}-{n}", 1, low_bit, False, round(load, 2), 0.0, False, True, test_api])
        return rows
    raise ValueError(f"unknown test_api {test_api}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config_path", default=os.path.join(os.path.dirname(__file__), "config.yaml"))
    ap.add_argument("-o", "--output_dir", default=".")
    a = ap.parse_args()
    conf = yaml.safe_load(open(a.config_path))
    today = date.today()
    results = []
    for api in conf["test_api"]:
        for model in conf["repo_id"]:
            bs = conf["batch_size"] if isinstance(conf["batch_size"], list) else [conf["batch_size"]]
            for b in bs:
                rows = run_model(model, api, conf["in_out_pairs"], conf.get("local_model_hub", ""), conf["warm_up"],
                                 conf["num_trials"], conf["low_bit"], b, conf)
                for r in rows:
                    r[-1] = api                  # the name the config asked for (an alias runs under its own name)
                results += rows
        out = os.path.join(a.output_dir, f"{api}-results-{today}.csv")
        rank0 = int(os.environ.get("RANK", 0)) == 0
        if rank0:
            with open(out, "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(COLUMNS)
                w.writerows([r for r in results if r[-1] == api])
            print(f"wrote {out}")
    for r in results:
        print(dict(zip(COLUMNS, r)))


if __name__ == "__main__":
    main()
