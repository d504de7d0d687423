"""Tensor-parallel serving: rank 0 runs the HTTP server and broadcasts each request; all ranks generate together
(reference: python/llm/example/GPU/Deepspeed-AutoTP-FastAPI/serving.py: ``dist.broadcast_object_list`` of the batch)."""
import argparse

import torch
import torch.distributed as dist
from transformers import AutoTokenizer

from bigdl_amd.parallel import init_distributed
from bigdl_amd.parallel.comm import staggered_wait
from bigdl_amd.parallel.tensor_parallel import tensor_parallel
from ipex_llm.transformers import AutoModelForCausalLM


def follower_loop(model, device):
    while True:
        box = [None]
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            break
        ids, kw = box[0]
        model._hf_generate(torch.tensor(ids, device=device), **kw)


class BroadcastingModel:
    """generate() wrapper for rank 0: ship the request to the followers, then run the same generate."""

    def __init__(self, model, device):
        self.model, self.device = model, device
        self.config, self.generation_config = model.config, model.generation_config

    def parameters(self):
        return self.model.parameters()

    def generate(self, input_ids, **kw):
        streamer = kw.pop("streamer", None)
        dist.broadcast_object_list([(input_ids.tolist(), kw)], src=0)
        return self.model._hf_generate(input_ids.to(self.device), streamer=streamer, **kw)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--port", type=int, default=8000)
    a = ap.parse_args()
    rank, world, device = init_distributed()
    staggered_wait()                                         # RANK_WAIT_TIME=<s>: ranks load one after another
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit, torch_dtype=torch.float16)
    model = tensor_parallel(model).half().to(device).eval()
    if rank == 0:
        import uvicorn
        from bigdl_amd.serving.fastapi import FastApp
        tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path)
        app = FastApp(BroadcastingModel(model, device), tok, served_model_name=a.repo_id_or_model_path)
        uvicorn.run(app.app, host="0.0.0.0", port=a.port)
        dist.broadcast_object_list([None], src=0)
    else:
        follower_loop(model, device)
