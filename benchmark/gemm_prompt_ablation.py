"""A/B timing of builds of the prompt GEMM in ONE process, rounds interleaved (separate processes differ by more than the
builds do: clocks, caches).  Every `libbigdl_hip_<name>.so` under bigdl_amd/lib/abl/ is loaded next to the production
library and called through its own handle; measurement builds (PROMPT_ABL, see csrc/hip/gemm_prompt_kernel.h) give
wrong results by construction, only their time matters.
    PYTHONPATH=. python benchmark/gemm_prompt_ablation.py [--shape qkv] [--cfg 1] [--libs abl1,abl2] [--rounds 12]"""
import argparse
import ctypes
import json
import os

import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.models.decoder import _random_t16
from bigdl_amd.ops import hip

SHAPES = {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008)}


def bind(path):
    L = ctypes.CDLL(path)
    vp, ci, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    L.bigdl_gemm_prompt.argtypes = [vp, vp, vp, vp, vp, vp, i64, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.bigdl_gemm_prompt_force.argtypes = [ci, ci]
    L.bigdl_gemm_prompt_force.restype = None
    return L


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="qkv")
    ap.add_argument("--cfg", type=int, default=-1)
    ap.add_argument("--m", type=int, default=1024)
    ap.add_argument("--splitk", type=int, default=0)
    ap.add_argument("--libs", default="")
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    abl = os.path.join(root, "bigdl_amd", "lib", "abl")
    found = sorted(f[len("libbigdl_hip_"):-3] for f in os.listdir(abl) if f.endswith(".so")) if os.path.isdir(abl) else []
    names = args.libs.split(",") if args.libs else found
    libs = {"production": bind(os.path.join(root, "bigdl_amd", "lib", "libbigdl_hip.so"))}
    for n in names:
        libs[n] = bind(os.path.join(abl, f"libbigdl_hip_{n}.so"))
    n, k = SHAPES[args.shape]
    m, qt = args.m, Q.SYM_INT4
    w = _random_t16(n, k, qt, torch.device("cuda"), torch.Generator(device="cuda").manual_seed(0), scale=0.02).data
    x = torch.randn(m, k, device="cuda", dtype=torch.float16)
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    ws = torch.empty(1 << 25, device="cuda", dtype=torch.float32)
    stream = torch.cuda.current_stream().cuda_stream

    def call(L):
        rc = L.bigdl_gemm_prompt(x.data_ptr(), w.data_ptr(), out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), m, n, k,
                                 x.stride(0), out.stride(0), qt, 0, 0, stream)
        assert rc == 0, rc

    for L in libs.values():
        L.bigdl_gemm_prompt_force(args.cfg, args.splitk)
        for _ in range(3):
            call(L)
    torch.cuda.synchronize()
    times = {name: [] for name in libs}
    for _ in range(args.rounds):
        for name, L in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                call(L)
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / args.iters * 1e3)
    for name, t in times.items():
        t = sorted(t)
        us = t[len(t) // 2]
        print(json.dumps({"build": name, "shape": args.shape, "cfg": args.cfg, "M": m, "median_us": round(us, 1), "min_us": round(t[0], 1),
                          "max_us": round(t[-1], 1), "tflops": round(2.0 * m * n * k / us / 1e6, 1)}), flush=True)
