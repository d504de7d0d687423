"""fp16 row-stream GEMV at the byte counts of the sym_int4 decode products (what a stream of that size costs in this form)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from bigdl_amd.ops import hip

def timeit(n, k, m, gu, form, copies=16):
    hip.gemv_f16_tune(rows_form=form)
    ws = [torch.randn(n, k, device="cuda", dtype=torch.float16) for _ in range(copies)]
    w2 = [torch.randn(n, k, device="cuda", dtype=torch.float16) for _ in range(copies)] if gu else None
    x = torch.randn(m, k, device="cuda", dtype=torch.float16)
    gamma = torch.ones(k, device="cuda", dtype=torch.float16)
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    def body():
        for i in range(copies):
            hip.gemv_f16(x, ws[i], w2=None if w2 is None else w2[i], gamma=gamma, act="silu" if gu else None, out=out)
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (20 * copies)
    nbytes = n * k * 2 * (2 if gu else 1)
    print(json.dumps({"N": n, "K": k, "M": m, "pair": gu, "form": form, "us": round(us, 2), "MB": round(nbytes / 1e6, 1),
                      "TB_per_s": round(nbytes / us / 1e6, 2)}), flush=True)

# byte counts of the int4 products: o 9.4 MB, qkv 28.3, down 25.4, gate/up 50.7
for n, k, gu in ((1152, 4096, False), (3456, 4096, False), (1152, 11008, False), (3104, 4096, True)):
    timeit(n, k, 1, gu, 1)
for m in (2, 3):
    for form in (1, 0):
        for n, k, gu in ((12288, 4096, False), (11008, 4096, True), (4096, 11008, False)):
            timeit(n, k, m, gu, form)
