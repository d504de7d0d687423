"""Padded (ragged) batch generation on the engine against an equal-length batch of the same size: the per-token cost
of the slot-mode decode and of the packed prompt pass next to the uniform-batch path.

    python benchmark/ragged_batch.py --model llama2-7b --batch 4 --len 256 --new 64 --out profiles/ragged.jsonl
"""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama2-7b")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--len", type=int, default=256)
    ap.add_argument("--new", type=int, default=64)
    ap.add_argument("--trials", type=int, default=3)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from bench import model_config
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import build_random_decoder
    cfg, _ = model_config(args.model)
    eng = build_random_decoder(cfg, Q.SYM_INT4, device="cuda", seed=0)
    b, s = args.batch, args.len
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(3, cfg.vocab_size, (b, s), generator=g).cuda()
    lens = [s if i in (0, b - 1) else max(8, int(s * f)) for i, f in zip(range(b), [1.0, 0.4, 0.7, 0.55, 0.9, 0.3, 0.8, 1.0] * 8)]
    mask = torch.zeros(b, s, dtype=torch.long, device="cuda")
    for i, n in enumerate(lens):
        mask[i, s - n:] = 1
    rows = []
    for name, kw in (("equal", {}), ("ragged_left_padded", dict(attention_mask=mask))):
        best = None
        for _ in range(args.trials + 1):
            eng.generate(ids, max_new_tokens=args.new, min_new_tokens=args.new, kv_fp8=True, **kw)
            cur = (eng.first_cost * 1e3, eng.rest_cost_mean * 1e3)
            best = cur if best is None or cur[1] < best[1] else best
        rows.append({"case": name, "batch": b, "prompt_lens": lens if kw else [s] * b, "first_ms": round(best[0], 3),
                     "ms_per_step": round(best[1], 4), "model": args.model})
        print(json.dumps(rows[-1]), flush=True)
    rows.append({"case": "ratio", "decode_ragged_over_equal": round(rows[1]["ms_per_step"] / rows[0]["ms_per_step"], 4),
                 "first_ragged_over_equal": round(rows[1]["first_ms"] / rows[0]["first_ms"], 4)})
    print(json.dumps(rows[-1]))
    if args.out:
        with open(args.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
