"""Pipeline-parallel generation: torchrun --standalone --nproc-per-node <stages> generate.py ...
(reference: python/llm/example/GPU/Pipeline-Parallel-Inference/generate.py)."""
import argparse

import torch
import torch.distributed as dist
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM, init_pipeline_parallel

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt", default="Once upon a time, there existed a little girl")
    ap.add_argument("--n-predict", type=int, default=32)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--gpu-num", type=int, default=2)
    a = ap.parse_args()
    init_pipeline_parallel()
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit, optimize_model=True,
                                                 torch_dtype=torch.float16, pipeline_parallel_stages=a.gpu_num)
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path)
    ids = tok(a.prompt, return_tensors="pt").input_ids.to(f"cuda:{dist.get_rank() % torch.cuda.device_count()}")
    out = model.generate(ids, max_new_tokens=a.n_predict)
    if dist.get_rank() == 0:
        print(tok.decode(out[0], skip_special_tokens=True))
        print(f"first token {model.first_token_time:.3f}s, rest {model.rest_cost_mean * 1e3:.2f} ms/token")
