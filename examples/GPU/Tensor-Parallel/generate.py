"""Tensor-parallel generation over RCCL: torchrun --standalone --nproc-per-node <N> generate.py ...
(reference: python/llm/example/GPU/Deepspeed-AutoTP/deepspeed_autotp.py -- there DeepSpeed AutoTP shards the model;
here the quantized weights are sharded directly, see bigdl_amd/parallel/tensor_parallel.py)."""
import argparse

import torch
from transformers import AutoTokenizer

from bigdl_amd.parallel import init_distributed
from bigdl_amd.parallel.comm import staggered_wait
from bigdl_amd.parallel.tensor_parallel import tensor_parallel
from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt", default="Once upon a time, there existed a little girl")
    ap.add_argument("--n-predict", type=int, default=32)
    ap.add_argument("--low-bit", default="sym_int4")
    a = ap.parse_args()
    rank, world, device = init_distributed()
    staggered_wait()                                         # RANK_WAIT_TIME=<s>: ranks load one after another
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit, optimize_model=True,
                                                 torch_dtype=torch.float16)
    model = tensor_parallel(model).half().to(device)        # shard on the host, then move this rank's shard
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path)
    ids = tok(a.prompt, return_tensors="pt").input_ids.to(device)
    out = model._hf_generate(ids, max_new_tokens=a.n_predict, do_sample=False)
    if rank == 0:
        print(tok.decode(out[0], skip_special_tokens=True))
