"""Long-running generation loop: checks that latency and memory stay flat over many requests.

Reference: python/llm/dev/benchmark/all-in-one/run-stress-test.py (driven by the nightly workflow
.github/workflows/llm_tests_for_stable_version_on_arc.yml:200-232).  Every round runs one generation of each in/out
pair; the script fails (exit 1) when the mean next-token latency of the last quarter of rounds exceeds that of the
first quarter by more than ``--tolerance`` or the peak memory keeps growing.

    python benchmark/stress/run_stress_test.py --model synthetic:llama2-7b --minutes 10
"""
from __future__ import annotations

import argparse
import csv
import json
import sys
import time


def drift(values, frac=0.25):
    """(mean of first quarter, mean of last quarter, relative change)"""
    n = max(1, int(len(values) * frac))
    a, b = sum(values[:n]) / n, sum(values[-n:]) / n
    return a, b, (b - a) / a if a else 0.0


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="synthetic:llama2-7b")
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--pairs", default="1024-128,2048-256")
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--max-rounds", type=int, default=100000)
    ap.add_argument("--tolerance", type=float, default=0.05)
    ap.add_argument("--csv", default="stress-results.csv")
    a = ap.parse_args()
    from bigdl_amd import qtypes as Q
    if a.model.startswith("synthetic:"):
        from bigdl_amd.models import decoder as D
        cfg = D.DecoderConfig(**{"llama2-7b": D.LLAMA2_7B, "tinyllama": D.TINY_LLAMA}[a.model.split(":", 1)[1]])
        eng = D.build_random_decoder(cfg, Q.ggml_tensor_qtype[a.low_bit])
        vocab = cfg.vocab_size
    else:
        from bigdl_amd.models.hf_patch import get_native_engine
        from ipex_llm.transformers import AutoModelForCausalLM
        model = AutoModelForCausalLM.from_pretrained(a.model, load_in_low_bit=a.low_bit).half().to("cuda")
        eng, vocab = get_native_engine(model), model.config.vocab_size
    pairs = [tuple(int(x) for x in p.split("-")) for p in a.pairs.split(",")]
    g = torch.Generator().manual_seed(0)
    rows, t_end = [], time.time() + a.minutes * 60
    rnd = 0
    while time.time() < t_end and rnd < a.max_rounds:
        for i, o in pairs:
            ids = torch.randint(3, vocab, (1, i), generator=g).cuda()
            eng.generate(ids, max_new_tokens=o, min_new_tokens=o, kv_fp8=True)
            rows.append([rnd, f"{i}-{o}", eng.first_cost * 1e3, eng.rest_cost_mean * 1e3,
                         torch.cuda.max_memory_reserved() / 2 ** 30])
        rnd += 1
        if rnd % 20 == 0:
            print(f"round {rnd}: {rows[-1]}", flush=True)
    with open(a.csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["round", "in_out", "1st token latency (ms)", "2+ avg latency (ms/token)", "peak mem (GB)"])
        w.writerows(rows)
    ok = True
    for p in {r[1] for r in rows}:
        lat = [r[3] for r in rows if r[1] == p]
        mem = [r[4] for r in rows if r[1] == p]
        a0, a1, d = drift(lat)
        grew = mem[-1] > mem[len(mem) // 2] * 1.01 if len(mem) > 4 else False
        print(json.dumps({"pair": p, "rounds": len(lat), "first_quarter_ms": round(a0, 4), "last_quarter_ms": round(a1, 4),
                          "drift": round(d, 4), "peak_mem_gb": round(max(mem), 2), "memory_still_growing": grew}))
        ok &= d <= a.tolerance and not grew
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
