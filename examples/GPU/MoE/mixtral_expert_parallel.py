"""Mixtral decode with expert parallelism on the native engine.

    torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 8 mixtral_expert_parallel.py \
        --repo-id-or-model-path <mixtral> [--mode a2a|allreduce]

``allreduce``: every rank sees the same prompt; each adds its experts' share and one all-reduce of [tokens, hidden] per
MoE layer sums them.  ``a2a``: every rank serves its own prompt (data-parallel attention); routed rows go to the rank
that owns the expert with an all-to-all in prefill and an all-gather / reduce-scatter pair inside the decode graph."""
import argparse

from transformers import AutoTokenizer

from bigdl_amd.models.hf_patch import get_native_engine
from bigdl_amd.parallel import init_distributed, shard_engine_experts
from ipex_llm.transformers import AutoModelForCausalLM

PROMPTS = ["The three primary colours are", "In one sentence, a mixture of experts is", "The capital of France is",
           "Briefly, xGMI links connect", "A haiku about memory bandwidth:", "The derivative of x squared is",
           "Once upon a time", "The fastest way to sort a list is"]

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--mode", default="a2a", choices=["a2a", "allreduce"])
    ap.add_argument("--n-predict", type=int, default=64)
    a = ap.parse_args()
    rank, world, device = init_distributed()
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_4bit=True).half().to(device)
    eng = shard_engine_experts(get_native_engine(model), rank, world, mode=a.mode)
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path)
    prompt = PROMPTS[rank % len(PROMPTS)] if a.mode == "a2a" else PROMPTS[0]
    ids = tok(prompt, return_tensors="pt").input_ids.to(device)
    # a2a mode: the ranks step together (collectives inside every MoE layer), so all of them decode n-predict tokens
    out = eng.generate(ids, max_new_tokens=a.n_predict, min_new_tokens=a.n_predict if a.mode == "a2a" else 0)
    if rank == 0 or a.mode == "a2a":
        print(f"[rank {rank}] {tok.decode(out[0], skip_special_tokens=True)!r}  {eng.rest_cost_mean * 1e3:.2f} ms/token")
