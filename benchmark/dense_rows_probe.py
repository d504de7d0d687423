"""Time of the dense fp16 GEMMs of a Llama-2-7B layer against the row count (tile quantization of the library GEMM):
what a prompt pass pays for extra rows that ride on it."""
import json
import sys

import torch

shapes = [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)]
ws = [torch.randn(n, k, device="cuda", dtype=torch.float16) * 0.02 for n, k in shapes]
rows = [int(r) for r in (sys.argv[1] if len(sys.argv) > 1 else "960,1024,1040,1088,1152,1280,2048,2112").split(",")]
for m in rows:
    xs = [torch.randn(m, k, device="cuda", dtype=torch.float16) for _, k in shapes]
    for _ in range(3):
        for x, w in zip(xs, ws):
            torch.nn.functional.linear(x, w)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        for x, w in zip(xs, ws):
            torch.nn.functional.linear(x, w)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    flops = 2 * m * sum(n * k for n, k in shapes)
    print(json.dumps({"rows": m, "layer_gemm_ms": round(ms, 4), "x32_layers_ms": round(ms * 32, 2),
                      "tflops": round(flops / ms / 1e9, 1)}), flush=True)
