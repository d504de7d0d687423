"""hipBLASLt fp8 (e4m3, row-wise scales) against fp16 GEMM at prompt shapes: what an opt-in fp8 prompt pass could gain.
    python benchmark/fp8_gemm_probe.py"""
import json

import torch


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
    f8 = torch.float8_e4m3fn
    for m in (1024, 8192):
        for name, (n, k) in {"qkv": (12288, 4096), "o": (4096, 4096), "gate": (11008, 4096), "down": (4096, 11008)}.items():
            x = torch.randn(m, k, device="cuda", dtype=torch.float16)
            w = torch.randn(n, k, device="cuda", dtype=torch.float16) * 0.02
            us16 = timed(lambda: torch.nn.functional.linear(x, w))
            rec = {"M": m, "shape": name, "fp16_us": round(us16, 1), "fp16_tflops": round(2.0 * m * n * k / us16 / 1e6, 1)}
            try:
                sx = x.abs().amax(1, keepdim=True).float() / 448.0
                sw = w.abs().amax(1, keepdim=True).float() / 448.0
                x8, w8 = (x / sx).to(f8), (w / sw).to(f8)
                fn = lambda: torch._scaled_mm(x8, w8.t(), scale_a=sx, scale_b=sw.t(), out_dtype=torch.bfloat16)
                y = fn()
                ref = torch.nn.functional.linear(x, w)
                rec["rel_err"] = round(((y.float() - ref.float()).abs().mean() / ref.float().abs().mean()).item(), 4)
                us8 = timed(fn)
                rec.update(fp8_us=round(us8, 1), fp8_tflops=round(2.0 * m * n * k / us8 / 1e6, 1))
                rec["quant_x_us"] = round(timed(lambda: (x / (x.abs().amax(1, keepdim=True).float() / 448.0)).to(f8)), 1)
            except Exception as e:      # noqa: BLE001
                rec["fp8_error"] = f"{type(e).__name__}: {str(e)[:160]}"
            print(json.dumps(rec), flush=True)
