"""LlamaIndex ``CustomLLM`` over a low-bit model (reference: python/llm/example/GPU/LlamaIndex/rag.py,
ipex_llm/llamaindex/llms/bigdlllm.py:90)."""
import argparse

from ipex_llm.llamaindex.llms import IpexLLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--question", default="What is AI?")
    a = ap.parse_args()
    llm = IpexLLM.from_model_id(model_name=a.model_path, tokenizer_name=a.model_path, context_window=2048,
                                max_new_tokens=128, generate_kwargs={"do_sample": False}, device_map="cuda")
    print(llm.complete(a.question).text)
    for chunk in llm.stream_complete(a.question):
        print(chunk.delta, end="", flush=True)
