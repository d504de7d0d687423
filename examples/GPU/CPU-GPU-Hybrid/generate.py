"""MoE model larger than the GPU: routed experts stay on the CPU as int4, everything else runs on the MI355X.
(reference: example/GPU/CPU-GPU-Hybrid-DeepSeek-R1-Inference/generate.py)

    python generate.py --repo-id-or-model-path /models/DeepSeek-R1 --gpu-layers 8 --prompt "..."
"""
import argparse
import time

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM
from bigdl_amd.transformers.hybrid import offload_to_cpu

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt", default="What is AI?")
    ap.add_argument("--n-predict", type=int, default=32)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--gpu-layers", type=int, default=0, help="experts of the first N decoder layers also go to the GPU")
    args = ap.parse_args()

    model = AutoModelForCausalLM.from_pretrained(args.repo_id_or_model_path, load_in_low_bit=args.low_bit,
                                                 optimize_model=True, trust_remote_code=True, use_cache=True)
    n = offload_to_cpu(model, skip_layers=range(args.gpu_layers))
    model = model.half().to("cuda")
    print(f"{n} expert containers stay in host memory")
    tok = AutoTokenizer.from_pretrained(args.repo_id_or_model_path, trust_remote_code=True)
    ids = tok(args.prompt, return_tensors="pt").input_ids.to("cuda")
    with torch.inference_mode():
        model.generate(ids, max_new_tokens=4)          # warm-up
        t0 = time.perf_counter()
        out = model.generate(ids, max_new_tokens=args.n_predict, do_sample=False)
        torch.cuda.synchronize()
    print(f"{time.perf_counter() - t0:.2f} s")
    print(tok.decode(out[0], skip_special_tokens=True))
