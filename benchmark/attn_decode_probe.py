"""Decode attention alone at a serving batch: ``--layers`` distinct e5m2 caches (so nothing is re-read from the
memory-side cache), one launch per cache inside a hipGraph, microseconds per launch and TB/s of K / V bytes.
    PYTHONPATH=. python benchmark/attn_decode_probe.py --batch 64 --ctx 1024 [--no-rope] [--ablate 1]
``--no-rope``: q already rotated and the new key / value already in the cache (the kernel without its append prologue).
``--ablate 1``: the kernel issues its K / V loads and skips the products (the load pattern's own rate)."""
import argparse
import ctypes
import json
import math

import torch

from bigdl_amd.ops import hip

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=32)
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--splits", type=int, default=1)
    ap.add_argument("--no-rope", action="store_true")
    ap.add_argument("--separate-rope", action="store_true", help="rope_append kernel + attention without the prologue")
    ap.add_argument("--ablate", type=int, default=0)
    args = ap.parse_args()
    b, hq, hkv, d = args.batch, args.heads, args.kv_heads, 128
    lmax = (args.ctx + 64 + 255) // 256 * 256
    dev = "cuda"
    if args.ablate:
        hip.lib().bigdl_attn_set_ablation.argtypes = [ctypes.c_int]
        hip.lib().bigdl_attn_set_ablation(args.ablate)
    kcs = [torch.randint(0, 120, (b, hkv, lmax, d), dtype=torch.uint8, device=dev) for _ in range(args.layers)]
    vcs = [torch.randint(0, 120, (b, hkv, d, lmax), dtype=torch.uint8, device=dev) for _ in range(args.layers)]
    qkv = (torch.randn(b, (hq + 2 * hkv) * d, device=dev) * 0.5).half()
    out = torch.empty(b, hq * d, dtype=torch.float16, device=dev)
    ws = torch.empty(hip.attn_decode_ws_floats(b, hq, hkv, 1, d, args.splits), dtype=torch.float32, device=dev)
    counters = torch.zeros(hip.attn_decode_counters(b, hq, hkv), dtype=torch.int32, device=dev)
    past = torch.full((b,), args.ctx, dtype=torch.int32, device=dev)
    half = d // 2
    ang = torch.arange(lmax, device=dev).float()[:, None] / (10000 ** (torch.arange(half, device=dev).float() / half))
    inv_freq = 1.0 / (10000 ** (torch.arange(half, device=dev).float() / half))
    table = torch.stack([ang.cos(), ang.sin()], -1).contiguous()

    def run():
        for kc, vc in zip(kcs, vcs):
            if args.separate_rope:
                hip.rope_append(qkv, kc, vc, past, inv_freq, n_heads=hq, n_kv_heads=hkv, head_dim=d, q_len=1, batch=b)
            hip.attn_decode(qkv, kc, vc, past, out, ws, counters, n_heads=hq, n_kv_heads=hkv, head_dim=d, q_len=1, batch=b,
                            scale=1.0 / math.sqrt(d), splits=args.splits, rope_table=None if args.no_rope or args.separate_rope else table)

    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(args.reps):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (args.reps * args.layers)
    nbytes = 2 * b * hkv * (args.ctx + 1) * d
    print(json.dumps({"batch": b, "heads": hq, "kv_heads": hkv, "ctx": args.ctx, "splits": args.splits, "rope_in_kernel": not (args.no_rope or args.separate_rope), "separate_rope": args.separate_rope,
                      "ablate": args.ablate, "us_per_launch": round(us, 2), "kv_tb_per_s": round(nbytes / us / 1e6, 3)}), flush=True)
