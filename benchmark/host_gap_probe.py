"""How much of the batch-1 decode step is host turnaround: 127 chained greedy steps (the token stays on the device)
replayed with a synchronize after every step, as ``generate`` does, against the same replays queued back to back.
    PYTHONPATH=. python benchmark/host_gap_probe.py"""
import json
import time

import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.models.decoder import LLAMA2_7B, DecoderConfig, build_random_decoder

if __name__ == "__main__":
    eng = build_random_decoder(DecoderConfig(**LLAMA2_7B), Q.SYM_INT4, device="cuda", seed=1)
    out = {}
    for mode in ("sync_each", "queued", "sync_each", "queued"):
        eng.ensure_cache(1, 1024 + 140, fp8=True)
        eng.cache.reset()
        ids = torch.randint(3, 32000, (1, 1024)).cuda()
        tok = eng.prefill(ids)[:, -1].argmax(-1).view(1, 1)
        eng.decode(tok, 1, chain=True)
        eng.decode(None, 1, chain=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(125):
            _, nxt = eng.decode(None, 1, chain=True)
            keep = nxt.clone()
            if mode == "sync_each":
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        out.setdefault(mode, []).append(round((time.perf_counter() - t0) / 125 * 1e3, 4))
    print(json.dumps({"ms_per_token": out}))
