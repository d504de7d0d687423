"""What a long prompt does to the sequences that are already running: inter-token gaps of 8 running requests while a
``--long``-token prompt is admitted whole or in chunks (``ContinuousBatcher.max_prefill_tokens``)."""
import argparse
import json
import time

import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.models.decoder import LLAMA2_7B, DecoderConfig, build_random_decoder
from bigdl_amd.serving.continuous import ContinuousBatcher, SeqRequest

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--long", type=int, default=16384)
    ap.add_argument("--chunks", default="0,4096,2048,1024")
    args = ap.parse_args()
    eng = build_random_decoder(DecoderConfig(**LLAMA2_7B), Q.SYM_INT4, device="cuda", seed=1)
    g = torch.Generator().manual_seed(0)
    short = [torch.randint(3, 32000, (128,), generator=g).tolist() for _ in range(8)]
    long_prompt = torch.randint(3, 32000, (args.long,), generator=g).tolist()
    for chunk in [int(c) for c in args.chunks.split(",")]:
        for rep in range(2):                                   # first repetition warms graphs and the allocator
            stamps = []
            cb = ContinuousBatcher(eng, max_num_seqs=9, max_seq_len=args.long + 256, kv_fp8=True,
                                   max_prefill_tokens=chunk or None, max_prefill_per_step=8)
            reqs = [cb.submit(SeqRequest(f"s{i}", p, max_new_tokens=96)) for i, p in enumerate(short)]
            for _ in range(16):
                cb.step()
            big = cb.submit(SeqRequest("long", long_prompt, max_new_tokens=8))
            torch.cuda.synchronize()
            stamps.append(time.perf_counter())
            while cb.has_work():
                cb.step()
                torch.cuda.synchronize()
                stamps.append(time.perf_counter())
            gaps = [b - a for a, b in zip(stamps, stamps[1:])]
        print(json.dumps({"long_prompt": args.long, "max_prefill_tokens": chunk or None,
                          "worst_gap_ms": round(max(gaps) * 1e3, 1),
                          "long_first_token_ms": round((big.first_token_time - big.arrival) * 1e3, 1),
                          "steps": cb.steps}), flush=True)
