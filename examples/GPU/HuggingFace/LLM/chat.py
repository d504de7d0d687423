"""Multi-turn streaming chat with any supported causal LM, with the prompt format of its family (reference:
python/llm/example/GPU/HuggingFace/LLM/<model>/generate.py, one directory per model; the formats below are what those
scripts hard-code).  The tokenizer's own chat template wins when it has one.

    python chat.py --repo-id-or-model-path <path> [--family llama2] [--low-bit sym_int4] [--kv fp8]
"""
import argparse
import os

import torch
from transformers import AutoTokenizer, TextStreamer

from ipex_llm.transformers import AutoModelForCausalLM

# family -> (system wrapper, one user turn, one assistant turn, generation prefix)
FORMATS = {
    "llama2": ("<<SYS>>\n{system}\n<</SYS>>\n\n", "[INST] {system}{content} [/INST]", " {content} </s><s>", ""),
    "llama3": ("<|start_header_id|>system<|end_header_id|>\n\n{system}<|eot_id|>", "<|start_header_id|>user<|end_header_id|>\n\n{content}<|eot_id|>",
               "<|start_header_id|>assistant<|end_header_id|>\n\n{content}<|eot_id|>", "<|start_header_id|>assistant<|end_header_id|>\n\n"),
    "mistral": ("", "[INST] {content} [/INST]", "{content}</s>", ""),
    "qwen": ("<|im_start|>system\n{system}<|im_end|>\n", "<|im_start|>user\n{content}<|im_end|>\n", "<|im_start|>assistant\n{content}<|im_end|>\n",
             "<|im_start|>assistant\n"),
    "chatglm3": ("<|system|>\n{system}", "<|user|>\n{content}", "<|assistant|>\n{content}", "<|assistant|>\n"),
    "baichuan2": ("{system}", "<reserved_106>{content}", "<reserved_107>{content}", "<reserved_107>"),
    "gemma": ("", "<start_of_turn>user\n{content}<end_of_turn>\n", "<start_of_turn>model\n{content}<end_of_turn>\n", "<start_of_turn>model\n"),
    "phi3": ("<|system|>\n{system}<|end|>\n", "<|user|>\n{content}<|end|>\n", "<|assistant|>\n{content}<|end|>\n", "<|assistant|>\n"),
    "yi": ("<|im_start|>system\n{system}<|im_end|>\n", "<|im_start|>user\n{content}<|im_end|>\n", "<|im_start|>assistant\n{content}<|im_end|>\n",
           "<|im_start|>assistant\n"),
    "internlm2": ("<|im_start|>system\n{system}<|im_end|>\n", "<|im_start|>user\n{content}<|im_end|>\n", "<|im_start|>assistant\n{content}<|im_end|>\n",
                  "<|im_start|>assistant\n"),
    "deepseek": ("{system}\n\n", "User: {content}\n\n", "Assistant: {content}<｜end▁of▁sentence｜>", "Assistant:"),
    "minicpm": ("", "<用户>{content}", "<AI>{content}", "<AI>"),
    "yuan2": ("", "{content}<sep>", "{content}<eod>", ""),
    "rwkv": ("", "User: {content}\n\n", "Assistant: {content}\n\n", "Assistant:"),
    "plain": ("", "{content}\n", "{content}\n", ""),
}


def render(family, system, turns):
    sys_fmt, user_fmt, bot_fmt, prefix = FORMATS[family]
    sys_text = sys_fmt.format(system=system) if system and sys_fmt else ""
    out, first = "", True
    for role, content in turns:
        if role == "user":
            if "{system}" in user_fmt:                       # llama2 puts the system block inside the first user turn
                out += user_fmt.format(system=sys_text if first else "", content=content)
            else:
                out += (sys_text if first else "") + user_fmt.format(content=content)
            first = False
        else:
            out += bot_fmt.format(content=content)
    return out + prefix


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--family", default=None, choices=sorted(FORMATS))
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--kv", default="fp16", choices=["fp16", "fp8", "int4"])
    ap.add_argument("--system", default="You are a helpful assistant.")
    ap.add_argument("--n-predict", type=int, default=256)
    ap.add_argument("--prompt", default=None, help="one-shot mode: answer this and exit")
    a = ap.parse_args()
    if a.kv != "fp16":
        os.environ["IPEX_LLM_QUANTIZE_KV_CACHE"] = "1" if a.kv == "fp8" else "int4"
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit, optimize_model=True,
                                                 trust_remote_code=True, use_cache=True)
    model = model.half().to(dev) if dev == "cuda" else model
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path, trust_remote_code=True)
    family = a.family or next((f for f in FORMATS if f in getattr(model.config, "model_type", "")), "plain")
    turns = []
    streamer = TextStreamer(tok, skip_prompt=True, skip_special_tokens=True)
    while True:
        text = a.prompt if a.prompt is not None else input("user> ")
        if not text or text in ("exit", "quit"):
            break
        turns.append(("user", text))
        if getattr(tok, "chat_template", None) and a.family is None:
            msgs = ([{"role": "system", "content": a.system}] if a.system else []) + \
                [{"role": "user" if r == "user" else "assistant", "content": c} for r, c in turns]
            prompt = tok.apply_chat_template(msgs, tokenize=False, add_generation_prompt=True)
        else:
            prompt = render(family, a.system, turns)
        ids = tok(prompt, return_tensors="pt").input_ids.to(dev)
        with torch.inference_mode():
            out = model.generate(ids, max_new_tokens=a.n_predict, do_sample=False, streamer=streamer)
        turns.append(("assistant", tok.decode(out[0, ids.shape[1]:], skip_special_tokens=True)))
        if a.prompt is not None:
            break
