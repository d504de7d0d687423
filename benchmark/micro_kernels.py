"""Kernel microbenchmarks on one MI355X: achieved HBM bandwidth of the fused low-bit GEMV on the
Llama-2-7B decode shapes, and decode-attention time at a 1K context.  Prints one JSON line per case."""
import json
import sys

import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.ops import hip, host


def timed_graph(fn, iters=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3  # us


def main():
    qt = Q.ggml_tensor_qtype[sys.argv[1]] if len(sys.argv) > 1 else Q.SYM_INT4
    shapes = {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (11008, 4096), "down": (4096, 11008),
              "lm_head": (32000, 4096)}
    # rotate over several weight copies so that successive iterations do not hit the 256 MiB L3
    for name, (n, k) in shapes.items():
        nbytes = host.device_nbytes(qt, n, k)
        copies = max(2, int(600e6 // nbytes))
        ws = [torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device="cuda") for _ in range(copies)]
        for m in (1, 4, 16):
            x = torch.randn(m, k, device="cuda").half()
            out = torch.empty(m, n, device="cuda", dtype=torch.float16)
            state = {"i": 0}

            def fn():
                w = ws[state["i"] % copies]
                state["i"] += 1
                if name == "gate_up":
                    hip.gemv_lowbit(x, w, qt, n, k, w2=ws[(state["i"]) % copies], act="silu", out=out)
                else:
                    hip.gemv_lowbit(x, w, qt, n, k, out=out)
            us = timed_graph(fn)
            total = nbytes * (2 if name == "gate_up" else 1)
            print(json.dumps({"kernel": "gemv", "qtype": Q.qtype_name(qt), "shape": name, "M": m, "us": round(us, 2),
                              "GBps": round(total / us / 1e3, 1)}), flush=True)
        del ws
    # decode attention, Llama-2-7B heads, fp16 and fp8 caches
    for fp8 in (False, True):
        for ctx in (1024, 4096):
            b, hq, hkv, d = 1, 32, 32, 128
            lmax = 8192
            dt = torch.uint8 if fp8 else torch.float16
            kc = torch.zeros(b, hkv, lmax, d, dtype=dt, device="cuda")
            vc = torch.zeros(b, hkv, d, lmax, dtype=dt, device="cuda")
            qkv = torch.randn(1, (hq + 2 * hkv) * d, device="cuda").half()
            out = torch.empty(1, hq * d, device="cuda", dtype=torch.float16)
            past = torch.full((1,), ctx, dtype=torch.int32, device="cuda")
            for splits in (1, 2, 4, 8):
                ws = torch.empty(hip.attn_decode_ws_floats(b, hq, hkv, 1, d, splits) + 16, device="cuda")
                cnt = torch.zeros(b * hkv, dtype=torch.int32, device="cuda")

                def fn():
                    hip.attn_decode(qkv, kc, vc, past, out, ws, cnt, n_heads=hq, n_kv_heads=hkv, head_dim=d, q_len=1,
                                    batch=b, scale=0.088, splits=splits)
                us = timed_graph(fn)
                kvbytes = 2 * hkv * ctx * d * (1 if fp8 else 2)
                print(json.dumps({"kernel": "attn_decode", "fp8": fp8, "ctx": ctx, "splits": splits,
                                  "us": round(us, 2), "GBps": round(kvbytes / us / 1e3, 1)}), flush=True)


if __name__ == "__main__":
    main()
