"""Achieved bandwidth of the fp16 skinny GEMM (csrc/hip/gemv_f16.hip) at the Llama-2-7B projection shapes, graph-timed
over 32 distinct weight copies so that nothing is served from cache."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from bigdl_amd.ops import hip  # noqa: E402

SHAPES = [("qkv", 12288, 4096, False), ("o", 4096, 4096, False), ("gate_up", 11008, 4096, True), ("down", 4096, 11008, False)]


def main():
    copies = 16
    # form 1: row-stream kernel for 1 - 4 rows (the default); form 0: the MFMA form for every row count
    for m, form in ((1, 1), (1, 0), (4, 1), (4, 0), (5, 1)):
        hip.gemv_f16_tune(rows_form=form)
        for name, n, k, gu in SHAPES:
            ws = [torch.randn(n, k, device="cuda", dtype=torch.float16) for _ in range(copies)]
            w2 = [torch.randn(n, k, device="cuda", dtype=torch.float16) for _ in range(copies)] if gu else None
            x = torch.randn(m, k, device="cuda", dtype=torch.float16)
            gamma = torch.ones(k, device="cuda", dtype=torch.float16)
            out = torch.empty(m, n, device="cuda", dtype=torch.float16)

            def body():
                for i in range(copies):
                    hip.gemv_f16(x, ws[i], w2=None if w2 is None else w2[i], gamma=gamma, act="silu" if gu else None, out=out)
            body()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (10 * copies)
            nbytes = n * k * 2 * (2 if gu else 1)
            print(json.dumps({"shape": name, "form": "rows" if form and m <= 4 else "mfma", "M": m, "N": n, "K": k, "us": round(us, 2), "MiB": round(nbytes / 2 ** 20, 1),
                              "TB_per_s": round(nbytes / us / 1e6, 2)}), flush=True)
            del ws, w2


if __name__ == "__main__":
    main()
