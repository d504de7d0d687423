"""Worker of tests/test_pipeline_gpu.py::test_engine_pipeline_serving: ranks of a gloo group share one GPU, each runs
the continuous-batching pipeline worker over its engine stage."""
import asyncio
import json
import os
import sys

import torch
import torch.distributed as dist


def main():
    out_path = sys.argv[1]
    stages = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import DecoderConfig, build_random_decoder
    from bigdl_amd.parallel.pipeline import EnginePPWorker, split_engine
    cfg = DecoderConfig(vocab_size=2048, hidden_size=512, intermediate_size=1536, num_layers=6, num_heads=8, num_kv_heads=4,
                        head_dim=64, max_position=2048)
    whole = build_random_decoder(cfg, Q.SYM_INT4, device="cuda:0", seed=7)
    g = torch.Generator().manual_seed(5)
    lens = [37, 12, 25, 18, 30]
    budgets = [9, 6, 12, 7, 8]
    prompts = [torch.randint(3, cfg.vocab_size, (n,), generator=g).tolist() for n in lens]
    res = {}
    if rank == 0:
        res["alone"] = [whole.generate(torch.tensor([p]), max_new_tokens=b)[0, len(p):].cpu().tolist()
                        for p, b in zip(prompts, budgets)]
    w = EnginePPWorker(split_engine(whole, rank, stages), rank, stages, max_num_seqs=4, max_prefilled_seqs=2, max_seq_len=256)

    async def drive():
        if rank == 0:
            for i, (p, b) in enumerate(zip(prompts, budgets)):
                await w.add_request(f"r{i}", p, b)
        out = await w.run_until_idle()
        w.shutdown()
        return out
    out = asyncio.run(drive())
    res["trace"] = w.trace
    res["prompt_passes"] = w.prompt_passes
    res["lanes_left"] = len(w.lanes)
    if rank == 0:
        res["served"] = {k: v.cpu().tolist() for k, v in out.items()}
        res["max_inflight"] = w.max_inflight
        streamed = {}
        for rid, q in w.streamer.items():
            items = []
            while not q.empty():
                items.append(q.get_nowait())
            streamed[rid] = items
        res["streamed"] = streamed
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
