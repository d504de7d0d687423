"""``llm-cli`` / ``llm-chat``: one-shot completion and interactive chat on a single-file model or a low-bit
checkpoint directory.  Reference: the bash / PowerShell launchers of ``main-<family>`` binaries,
python/llm/src/ipex_llm/cli/llm-cli:25-57 and llm-chat."""
from __future__ import annotations

import argparse
import os
import sys


def _load(path, family, low_bit):
    import torch
    if os.path.isdir(path):
        from ..transformers import AutoModelForCausalLM
        from transformers import AutoTokenizer
        try:
            model = AutoModelForCausalLM.load_low_bit(path)
        except Exception:
            model = AutoModelForCausalLM.from_pretrained(path, load_in_low_bit=low_bit)
        tok = AutoTokenizer.from_pretrained(path)
    else:
        from ..transformers import AutoModelForCausalLM
        model, tok = AutoModelForCausalLM.from_gguf(path, low_bit=low_bit)
    if torch.cuda.is_available():
        model = model.half().to("cuda")
    return model, tok


def _complete(model, tok, prompt, n_predict, temperature, stream=True):
    import torch
    from transformers import TextStreamer
    ids = tok(prompt, return_tensors="pt").input_ids.to(next(model.parameters()).device)
    streamer = TextStreamer(tok, skip_prompt=True, skip_special_tokens=True) if stream else None
    kw = dict(do_sample=True, temperature=temperature) if temperature and temperature > 0 else dict(do_sample=False)
    with torch.inference_mode():
        out = model.generate(ids, max_new_tokens=n_predict, streamer=streamer, **kw)
    return tok.decode(out[0, ids.shape[1]:], skip_special_tokens=True)


def _parser(desc):
    ap = argparse.ArgumentParser(description=desc)
    ap.add_argument("-m", "--model", required=True, help="model file (.gguf) or low-bit checkpoint directory")
    ap.add_argument("-x", "--model_family", "--model-family", default="llama")
    ap.add_argument("-t", "--threads", type=int, default=0)
    ap.add_argument("-n", "--n_predict", "--n-predict", type=int, default=128)
    ap.add_argument("--temp", type=float, default=0.0)
    ap.add_argument("--low-bit", default="sym_int4")
    return ap


def main(argv=None):
    ap = _parser("llm-cli: run a prompt through a low-bit model")
    ap.add_argument("-p", "--prompt", required=True)
    a = ap.parse_args(argv)
    if a.threads:
        import torch
        torch.set_num_threads(a.threads)
    model, tok = _load(a.model, a.model_family, a.low_bit)
    _complete(model, tok, a.prompt, a.n_predict, a.temp)
    print()


def chat_main(argv=None):
    ap = _parser("llm-chat: interactive chat with a low-bit model")
    ap.add_argument("--system", default="")
    a = ap.parse_args(argv)
    model, tok = _load(a.model, a.model_family, a.low_bit)
    history = [{"role": "system", "content": a.system}] if a.system else []
    print("Type your message (Ctrl-D or /exit to quit).")
    while True:
        try:
            user = input("> ").strip()
        except EOFError:
            break
        if user in ("/exit", "/quit"):
            break
        if not user:
            continue
        history.append({"role": "user", "content": user})
        if getattr(tok, "chat_template", None):
            prompt = tok.apply_chat_template(history, tokenize=False, add_generation_prompt=True)
        else:
            prompt = "".join(f"{m['role'].upper()}: {m['content']}\n" for m in history) + "ASSISTANT:"
        reply = _complete(model, tok, prompt, a.n_predict, a.temp)
        print()
        history.append({"role": "assistant", "content": reply})


if __name__ == "__main__":
    sys.exit(main())
