"""Worker of tests/test_pipeline_gpu.py: ranks of a gloo group share one GPU, each runs one engine stage."""
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
    from bigdl_amd.parallel.pipeline import EnginePipeline, split_engine
    cfg = DecoderConfig(vocab_size=2048, hidden_size=512, intermediate_size=1536, num_layers=6, num_heads=8, num_kv_heads=4,
                        head_dim=64, max_position=2048)
    whole = build_random_decoder(cfg, Q.SYM_INT4, device="cuda:0", seed=7)
    g = torch.Generator().manual_seed(3)
    prompt = torch.randint(3, cfg.vocab_size, (6, 37), generator=g)
    res = {}
    if rank == 0:
        res["whole"] = whole.generate(prompt, max_new_tokens=12).cpu().tolist()
        # (micro-batches run the kernels at their own row count: compare them with the whole engine at that row count)
        res["whole_mb"] = torch.cat([whole.generate(c, max_new_tokens=12) for c in prompt.chunk(stages)]).cpu().tolist()
    pipe = EnginePipeline(split_engine(whole, rank, stages))
    res["pipe"] = pipe.generate(prompt, max_new_tokens=12).cpu().tolist()
    res["pipe_mb"] = pipe.generate(prompt, max_new_tokens=12, micro_batches=stages).cpu().tolist()
    res["ticks"] = pipe.ticks_busy
    # early stop: the token the whole engine emits at step 5 of sequence 0 becomes EOS for a batch-1 run
    one = prompt[:1]
    ref = torch.tensor(res["pipe"])[0, 37:]
    eos = int(ref[5])
    res["eos"] = eos
    res["stopped"] = pipe.generate(one, max_new_tokens=12, eos_token_id=eos).cpu().tolist()
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
