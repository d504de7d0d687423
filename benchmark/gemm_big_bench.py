"""The prompt-sized fused low-bit GEMM (gemm_prompt) against dequantize + hipBLASLt and hipBLASLt alone on the
Llama-2-7B projection shapes; also checks the result against the dequantized product.
    PYTHONPATH=. python benchmark/gemm_big_bench.py [--ms 512,1024,2048] [--qtype sym_int4] [--sweep]"""
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
    ap.add_argument("--sweep", action="store_true", help="every tile configuration / K split instead of the planned one")
    ap.add_argument("--no-ref", action="store_true", help="skip the library comparisons")
    args = ap.parse_args()
    qt = Q.ggml_tensor_qtype[args.qtype]
    gen = torch.Generator(device="cuda").manual_seed(0)
    for name, (n, k) in SHAPES.items():
        w = _random_t16(n, k, qt, torch.device("cuda"), gen, scale=0.02).data
        wd = hip.dequant_t16(w, qt, n, k)
        for m in [int(v) for v in args.ms.split(",")]:
            x = torch.randn(m, k, device="cuda", dtype=torch.float16)
            ref = torch.nn.functional.linear(x.float(), wd.float())
            forces = [(-1, 0)]
            if args.sweep:
                forces += [(c, sk) for c in (0, 1, 2) for sk in (1, 2, 4) if sk == 1 or k // 128 // sk >= 8]
            us_dense = us_blas = None
            if not args.no_ref:
                us_dense = timed(lambda: torch.nn.functional.linear(x, hip.dequant_t16(w, qt, n, k)))
                us_blas = timed(lambda: torch.nn.functional.linear(x, wd))
            for cfg, sk in forces:
                hip.gemm_prompt_force(cfg, sk)
                got = hip.gemm_prompt(x, w, qt, n, k).float()
                err = ((got - ref).abs().max() / ref.abs().max()).item()
                us = timed(lambda: hip.gemm_prompt(x, w, qt, n, k))
                plan = hip.gemm_prompt_plan(m, n, k)
                hip.gemm_prompt_force()
                print(json.dumps({"qtype": args.qtype, "shape": name, "M": m, "N": n, "K": k, "forced": cfg >= 0,
                                  "cfg": plan[0], "splitk": plan[1], "max_rel_err": round(err, 5),
                                  "fused_us": round(us, 1), "fused_tflops": round(2.0 * m * n * k / us / 1e6, 1),
                                  "dequant_plus_hipblaslt_us": us_dense and round(us_dense, 1),
                                  "hipblaslt_only_us": us_blas and round(us_blas, 1)}), flush=True)
