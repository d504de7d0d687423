"""Batched offline generation with low-bit weights: vLLM (ROCm build) when it is installed, otherwise this package's own
batching engine behind the same interface (reference: python/llm/example/GPU/vLLM-Serving/offline_inference.py).
Server: python -m ipex_llm.vllm.xpu.entrypoints.openai.api_server --model <path> --load-in-low-bit fp8 --tensor-parallel-size 8"""
import argparse

try:
    from vllm import SamplingParams
except ImportError:                      # no vLLM: the same interface on this package's continuous batcher
    from ipex_llm.vllm import SamplingParams

from ipex_llm.vllm.xpu.engine import IPEXLLMClass as LLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True)
    ap.add_argument("--load-in-low-bit", default="sym_int4")
    ap.add_argument("--tensor-parallel-size", type=int, default=1)
    a = ap.parse_args()
    llm = LLM(model=a.model, dtype="float16", enforce_eager=True, load_in_low_bit=a.load_in_low_bit,
              tensor_parallel_size=a.tensor_parallel_size, trust_remote_code=True)
    prompts = ["Hello, my name is", "The president of the United States is", "The capital of France is", "The future of AI is"]
    for out in llm.generate(prompts, SamplingParams(temperature=0.8, top_p=0.95)):
        print(f"Prompt: {out.prompt!r}, Generated text: {out.outputs[0].text!r}")
