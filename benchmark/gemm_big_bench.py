"""The prompt-sized fused low-bit GEMM (gemm_lowbit_big) against dequantize + hipBLASLt and hipBLASLt alone on the
Llama-2-7B projection shapes; also checks the result against the dequantized product.
    PYTHONPATH=. python benchmark/gemm_big_bench.py [--ms 512,1024,2048] [--qtype sym_int4]"""
import argparse
import json

import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.models.decoder import _random_t16
from bigdl_amd.ops import hip

SHAPES = {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008)}


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


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--qtype", default="sym_int4")
    ap.add_argument("--ms", default="1024")
    args = ap.parse_args()
    qt = Q.ggml_tensor_qtype[args.qtype]
    gen = torch.Generator(device="cuda").manual_seed(0)
    for name, (n, k) in SHAPES.items():
        w = _random_t16(n, k, qt, torch.device("cuda"), gen, scale=0.02).data
        wd = hip.dequant_t16(w, qt, n, k)
        for m in [int(v) for v in args.ms.split(",")]:
            x = torch.randn(m, k, device="cuda", dtype=torch.float16)
            ref = torch.nn.functional.linear(x.float(), wd.float())
            got = hip.gemm_lowbit_big(x, w, qt, n, k).float()
            err = ((got - ref).abs().max() / ref.abs().max()).item()
            us_big = timed(lambda: hip.gemm_lowbit_big(x, w, qt, n, k))
            us_small = timed(lambda: hip.gemm_lowbit(x, w, qt, n, k)) if m <= 256 else None
            us_dense = timed(lambda: torch.nn.functional.linear(x, hip.dequant_t16(w, qt, n, k)))
            us_blas = timed(lambda: torch.nn.functional.linear(x, wd))
            print(json.dumps({"qtype": args.qtype, "shape": name, "M": m, "N": n, "K": k, "max_rel_err": round(err, 5),
                              "splitk": hip.lib().bigdl_gemm_lowbit_big_splitk(m, n, k),
                              "fused_big_us": round(us_big, 1), "fused_rows_le_256_us": us_small and round(us_small, 1), "dequant_plus_hipblaslt_us": round(us_dense, 1),
                              "hipblaslt_only_us": round(us_blas, 1),
                              "fused_big_tflops": round(2.0 * m * n * k / us_big / 1e6, 1),
                              "dequant_plus_hipblaslt_tflops": round(2.0 * m * n * k / us_dense / 1e6, 1)}), flush=True)
