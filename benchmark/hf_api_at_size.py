"""The public HuggingFace-style route at production size, timed end to end:

    random-init model of a named shape (from_config) -> optimize_model(low_bit) -> save_low_bit
    -> AutoModelForCausalLM.load_low_bit -> .to('cuda') -> BenchmarkWrapper(model).generate

which is the ``transformer_int4_fp16_loadlowbit_gpu`` mode of the reference's all-in-one benchmark
(python/llm/dev/benchmark/all-in-one/run.py:1330-1420; load + ``.to('xpu')`` timing :1345-1360, the timed generate loop
:1375-1410), followed by the native engine built directly on random low-bit blocks of the same shape for comparison.
One JSON line per stage.  No checkpoint exists on the box, so the weights are random (drawn on the GPU when there
is one, then moved to the host, because a 7B-parameter normal_() on the host takes minutes).

    python benchmark/hf_api_at_size.py --model llama2-7b --low-bit sym_int4 --in-len 1024 --out-len 128
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

SHAPES = {
    "llama2-7b": dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                      num_attention_heads=32, num_key_value_heads=32, max_position_embeddings=4096, rms_norm_eps=1e-5),
    "tiny": dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                 num_key_value_heads=2, max_position_embeddings=2048),
}


def log(stage, **kw):
    print(json.dumps({"stage": stage, **kw}), flush=True)


def random_model(name, gen_device):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(**SHAPES[name], tie_word_embeddings=False)
    with torch.device("meta"):
        model = LlamaForCausalLM(cfg)
    model = model.to_empty(device="cpu").half()
    g = torch.Generator(device=gen_device).manual_seed(0)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, device=gen_device, generator=g, dtype=torch.float32) * 0.02).half())
        # buffers created on the meta device (rotary inv_freq) are uninitialised after to_empty: take them from a
        # freshly constructed module
        for m in model.modules():
            if hasattr(m, "inv_freq"):
                fresh = type(m)(cfg)
                m.inv_freq = fresh.inv_freq.float()
                if hasattr(m, "original_inv_freq"):
                    m.original_inv_freq = fresh.inv_freq.float()
    return model.eval()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama2-7b", choices=list(SHAPES))
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--in-len", type=int, default=1024)
    ap.add_argument("--out-len", type=int, default=128)
    ap.add_argument("--trials", type=int, default=3)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--cpu", action="store_true", help="stop after load_low_bit + a short generate on the host")
    a = ap.parse_args()
    cuda = torch.cuda.is_available() and not a.cpu
    from bigdl_amd.optimize import optimize_model
    from bigdl_amd.transformers import AutoModelForCausalLM
    from bigdl_amd.utils.benchmark_util import BenchmarkWrapper
    work = a.dir or tempfile.mkdtemp(prefix="bigdl_lowbit_")
    try:
        t0 = time.perf_counter()
        model = random_model(a.model, "cuda" if cuda else "cpu")
        log("from_config_random_init", seconds=round(time.perf_counter() - t0, 2),
            params_b=round(sum(p.numel() for p in model.parameters()) / 1e9, 3))
        t0 = time.perf_counter()
        model = optimize_model(model, low_bit=a.low_bit)
        log("optimize_model", low_bit=a.low_bit, seconds=round(time.perf_counter() - t0, 2))
        t0 = time.perf_counter()
        model.save_low_bit(work)
        size = sum(os.path.getsize(os.path.join(work, f)) for f in os.listdir(work))
        log("save_low_bit", seconds=round(time.perf_counter() - t0, 2), gib=round(size / 2 ** 30, 3))
        del model
        t0 = time.perf_counter()
        model = AutoModelForCausalLM.load_low_bit(work, torch_dtype=torch.float16 if cuda else torch.float32)
        t_load = time.perf_counter() - t0
        log("load_low_bit", seconds=round(t_load, 2))
        g = torch.Generator().manual_seed(0)
        ids = torch.randint(3, SHAPES[a.model]["vocab_size"], (1, a.in_len), generator=g)
        if not cuda:
            out = model.generate(ids[:, :16], max_new_tokens=4, do_sample=False)
            log("cpu_generate", tokens=out.shape[1] - 16)
            return
        t0 = time.perf_counter()
        model = model.half().to("cuda")
        torch.cuda.synchronize()
        log("to_cuda", seconds=round(time.perf_counter() - t0, 2), note="host re-layout into device tiles + upload")
        wrapped = BenchmarkWrapper(model, do_print=False)
        ids = ids.cuda()
        firsts, rests = [], []
        for t in range(a.trials + 1):
            t0 = time.perf_counter()
            out = wrapped.generate(ids, max_new_tokens=a.out_len, min_new_tokens=a.out_len, do_sample=False)
            if t == 0:
                log("first_generate_call", seconds=round(time.perf_counter() - t0, 2),
                    note="engine build (zero-copy views of the module weights) + graph capture + generation")
            else:
                firsts.append(wrapped.first_cost)
                rests.append(wrapped.rest_cost_mean)
        api_rest = sum(rests) / len(rests)
        log("transformer_int4_fp16_loadlowbit_gpu", first_token_ms=round(1e3 * sum(firsts) / len(firsts), 3),
            next_token_ms=round(1e3 * api_rest, 4), peak_mem_gb=round(wrapped.peak_memory, 2), new_tokens=out.shape[1] - a.in_len)
        del wrapped, model
        torch.cuda.empty_cache()
        # the engine on random blocks of the same shape (what bench.py times)
        from bigdl_amd import qtypes as Q
        from bigdl_amd.models import decoder as D
        s = SHAPES[a.model]
        cfg = D.DecoderConfig(vocab_size=s["vocab_size"], hidden_size=s["hidden_size"], intermediate_size=s["intermediate_size"],
                              num_layers=s["num_hidden_layers"], num_heads=s["num_attention_heads"],
                              num_kv_heads=s["num_key_value_heads"], head_dim=s["hidden_size"] // s["num_attention_heads"],
                              rms_eps=s.get("rms_norm_eps", 1e-6), max_position=s["max_position_embeddings"])
        eng = D.build_random_decoder(cfg, Q.ggml_tensor_qtype[a.low_bit], device="cuda", seed=1)
        from bigdl_amd.transformers.utils import engine_kv_mode
        rests, firsts = [], []
        for t in range(a.trials + 1):
            eng.generate(ids, max_new_tokens=a.out_len, kv_fp8=engine_kv_mode())
            if t:
                firsts.append(eng.first_cost)
                rests.append(eng.rest_cost_mean)
        eng_rest = sum(rests) / len(rests)
        log("native_engine_gpu", first_token_ms=round(1e3 * sum(firsts) / len(firsts), 3), next_token_ms=round(1e3 * eng_rest, 4),
            api_over_engine=round(api_rest / eng_rest, 4))
    finally:
        if a.dir is None:
            shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
