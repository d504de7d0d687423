"""Load from the ModelScope hub instead of the HuggingFace hub (``model_hub="modelscope"``; needs the ``modelscope``
package; reference: python/llm/example/GPU/ModelScope-Models/)."""
import argparse

import torch

from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", default="qwen/Qwen2-7B-Instruct")
    a = ap.parse_args()
    from modelscope import AutoTokenizer
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_4bit=True, model_hub="modelscope",
                                                 trust_remote_code=True).half().to("cuda")
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path, trust_remote_code=True)
    ids = tok("What is AI?", return_tensors="pt").input_ids.to("cuda")
    with torch.inference_mode():
        print(tok.decode(model.generate(ids, max_new_tokens=32)[0], skip_special_tokens=True))
