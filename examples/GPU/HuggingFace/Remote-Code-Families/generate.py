"""ChatGLM2/3, GLM-4, Baichuan(2)-7B, Qwen (v1), InternLM(2), MiniCPM, Aquila: checkpoints whose modeling code only exists
as remote code are loaded through their weight-name maps into the native engine -- no ``trust_remote_code`` execution
(reference examples: python/llm/example/GPU/HuggingFace/LLM/{chatglm3,baichuan2,qwen,internlm2,minicpm}/generate.py)."""
import argparse

from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt", default="AI是什么？")
    ap.add_argument("--n-predict", type=int, default=32)
    ap.add_argument("--low-bit", default="sym_int4")
    a = ap.parse_args()
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit).to("cuda")
    # only the tokenizer still comes from the checkpoint's own python files
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path, trust_remote_code=True)
    ids = tok(a.prompt, return_tensors="pt").input_ids.to("cuda")
    out = model.generate(ids, max_new_tokens=a.n_predict, eos_token_id=tok.eos_token_id)
    print(tok.decode(out[0], skip_special_tokens=True))
    print(f"first token {model.first_cost * 1e3:.1f} ms, next tokens {model.rest_cost_mean * 1e3:.2f} ms/token")
    model.save_low_bit(a.repo_id_or_model_path.rstrip("/") + "-" + a.low_bit)     # reload with load_low_bit(...)
