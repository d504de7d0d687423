"""Pipeline-parallel generation on the native engine: every rank holds a contiguous range of layers and replays one
hipGraph per token; with --micro-batches equal to the number of stages every GPU is busy at every tick.

    torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 4 engine_pipeline.py --repo-id-or-model-path <path> \
        --batch 8 --micro-batches 4
"""
import argparse
import time

import torch
import torch.distributed as dist
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM, init_pipeline_parallel

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--prompt", default="Once upon a time")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--micro-batches", type=int, default=0, help="0: one per stage")
    ap.add_argument("--n-predict", type=int, default=64)
    a = ap.parse_args()
    init_pipeline_parallel()
    rank, world = dist.get_rank(), dist.get_world_size()
    from bigdl_amd.models.hf_patch import get_native_engine
    from bigdl_amd.parallel.pipeline import EnginePipeline, split_engine
    # every rank converts the checkpoint on the host, keeps its own layers on its GPU and frees the rest
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit, optimize_model=True)
    start = model.config.num_hidden_layers // world * rank
    end = model.config.num_hidden_layers if rank == world - 1 else start + model.config.num_hidden_layers // world
    for i, layer in enumerate(model.model.layers):
        if start <= i < end:
            layer.half().to("cuda")
    model.model.embed_tokens.half().to("cuda")
    model.model.norm.half().to("cuda")
    model.lm_head.half().to("cuda")
    # engine over the local layers only (the list is cut before the engine is built, so nothing else reaches the GPU)
    model.model.layers = torch.nn.ModuleList(list(model.model.layers)[start:end])
    model.config.num_hidden_layers = end - start
    from bigdl_amd.parallel.pipeline import make_stage
    pipe = EnginePipeline(make_stage(get_native_engine(model), rank, world))
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path)
    ids = tok([a.prompt] * a.batch, return_tensors="pt").input_ids
    for _ in range(2):                                  # first call captures the graphs
        t0 = time.perf_counter()
        out = pipe.generate(ids, max_new_tokens=a.n_predict, eos_token_id=tok.eos_token_id,
                            micro_batches=a.micro_batches or world)
        dt = time.perf_counter() - t0
    if rank == 0:
        print(tok.decode(out[0], skip_special_tokens=True))
        new = (out.shape[1] - ids.shape[1]) * a.batch
        print(f"-- {new} tokens in {dt:.2f} s ({new / dt:.0f} tokens/s), first token {pipe.first_cost * 1e3:.1f} ms, "
              f"{pipe.rest_cost_mean * 1e3:.2f} ms per step of {a.batch} sequences")
    dist.barrier()
