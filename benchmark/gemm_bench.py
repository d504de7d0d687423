"""Fused low-bit GEMM vs dequantize + dense GEMM (hipBLASLt) over the Llama-2-7B projection shapes.

    python benchmark/gemm_bench.py [--qtype sym_int4] > gemm.jsonl
"""
import argparse
import json

import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.models.decoder import _random_t16
from bigdl_amd.ops import hip

SHAPES = {"qkv": (12288, 4096), "o": (4096, 4096), "gate": (11008, 4096), "down": (4096, 11008)}


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--qtype", default="sym_int4")
    ap.add_argument("--ms", default="32,64,96,128,192,256,384,512,1024")
    args = ap.parse_args()
    qt = Q.ggml_tensor_qtype[args.qtype]
    gen = torch.Generator(device="cuda").manual_seed(0)
    for name, (n, k) in SHAPES.items():
        w = _random_t16(n, k, qt, torch.device("cuda"), gen, scale=0.02).data
        for m in [int(v) for v in args.ms.split(",")]:
            x = torch.randn(m, k, device="cuda", dtype=torch.float16)
            us_fused = timed(lambda: hip.gemm_lowbit(x, w, qt, n, k))
            us_dense = timed(lambda: torch.nn.functional.linear(x, hip.dequant_t16(w, qt, n, k)))
            wd = hip.dequant_t16(w, qt, n, k)
            us_blas = timed(lambda: torch.nn.functional.linear(x, wd))
            rec = {"kernel": "gemm_lowbit", "qtype": args.qtype, "shape": name, "M": m, "N": n, "K": k,
                   "fused_us": round(us_fused, 2), "dequant_plus_hipblaslt_us": round(us_dense, 2),
                   "hipblaslt_only_us": round(us_blas, 2),
                   "fused_tflops": round(2.0 * m * n * k / us_fused / 1e6, 1)}
            if m <= 64:
                rec["gemv_us"] = round(timed(lambda: hip.gemv_lowbit(x, w, qt, n, k)), 2)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
