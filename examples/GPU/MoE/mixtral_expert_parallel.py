"""Mixtral decode with expert parallelism on the native engine: every rank holds all attention weights and
E / world experts; one all-reduce of [tokens, hidden] per MoE layer replaces the dispatch / return all-to-all pair.
    torchrun --standalone --nproc-per-node 8 mixtral_expert_parallel.py --repo-id-or-model-path <mixtral>"""
import argparse

import torch
from transformers import AutoTokenizer

from bigdl_amd.models.hf_patch import get_native_engine
from bigdl_amd.parallel import init_distributed
from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt", default="The three primary colours are")
    ap.add_argument("--n-predict", type=int, default=64)
    a = ap.parse_args()
    rank, world, device = init_distributed()
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_4bit=True).half().to(device)
    eng = get_native_engine(model)
    for lw in eng.layers:
        if lw.moe is not None:
            lw.moe.shard_experts(rank, world)
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path)
    ids = tok(a.prompt, return_tensors="pt").input_ids.to(device)
    out = eng.generate(ids, max_new_tokens=a.n_predict)
    if rank == 0:
        print(tok.decode(out[0], skip_special_tokens=True))
        print(f"{eng.rest_cost_mean * 1e3:.2f} ms/token")
