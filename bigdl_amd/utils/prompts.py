"""Prompt formats of the model families the reference ships one example directory for
(python/llm/example/GPU/HuggingFace/LLM/<family>/generate.py: each script carries its own ``*_PROMPT_FORMAT`` string or
``get_prompt`` function).  Here they are one table, so that a single script (examples/GPU/HuggingFace/LLM/generate.py),
the chat example, the serving front ends and the benchmarks build a family's prompt the same way.

    from bigdl_amd.utils.prompts import build_prompt
    text = build_prompt("llama2", "What is AI?")
    text = build_prompt("llama3", "And in one word?", history=[("What is AI?", "...")], system="Be brief.")

Three kinds of entries:
  * a single-turn template with a ``{prompt}`` slot (most families);
  * a multi-turn builder (llama2 / llama3 / ChatML families: system prompt and history are part of the format);
  * ``CHAT_TEMPLATE``: the family's tokenizer carries the format (``tokenizer.apply_chat_template``); ``build_prompt``
    needs the tokenizer then.
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional, Sequence, Tuple, Union

History = Sequence[Tuple[str, str]]
CHAT_TEMPLATE = "<tokenizer chat template>"


def _llama2(prompt: str, history: History = (), system: str = "") -> str:
    """[INST] <<SYS>> ... <</SYS>> first turn, then alternating [/INST] answer </s><s>[INST] turns.  The first user
    turn is kept as typed, later ones are stripped (the convention of the Llama-2 chat checkpoints)."""
    out = [f"<s>[INST] <<SYS>>\n{system}\n<</SYS>>\n\n"]
    for i, (user, answer) in enumerate(history):
        user = user.strip() if i else user
        out.append(f"{user} [/INST] {answer.strip()} </s><s>[INST] ")
    out.append(f"{prompt.strip() if history else prompt} [/INST]")
    return "".join(out)


def _llama3(prompt: str, history: History = (), system: str = "") -> str:
    out = ["<|begin_of_text|>"]
    if system:
        out.append(f"<|start_header_id|>system<|end_header_id|>\n\n{system}<|eot_id|>")
    for user, answer in history:
        out.append(f"<|start_header_id|>user<|end_header_id|>\n\n{user}<|eot_id|>")
        out.append(f"<|start_header_id|>assistant<|end_header_id|>\n\n{answer}<|eot_id|>")
    out.append(f"<|start_header_id|>user<|end_header_id|>\n\n{prompt}<|eot_id|>")
    out.append("<|start_header_id|>assistant<|end_header_id|>\n\n")
    return "".join(out)


def _chatml(default_system: str = "") -> Callable[..., str]:
    def build(prompt: str, history: History = (), system: str = "") -> str:
        system = system or default_system
        out = [f"<|im_start|>system\n{system}<|im_end|>\n"] if system else []
        for user, answer in history:
            out.append(f"<|im_start|>user\n{user}<|im_end|>\n<|im_start|>assistant\n{answer}<|im_end|>\n")
        out.append(f"<|im_start|>user\n{prompt}<|im_end|>\n<|im_start|>assistant\n")
        return "".join(out)
    return build


def _rwkv(sep: str) -> Callable[..., str]:
    """RWKV world models: a greeting turn first, turns separated by ``sep`` (blank line for v4, newline for v5);
    newlines inside an instruction are collapsed because a blank line ends a turn."""
    greeting = ("Hi. I am your assistant and I will provide expert full response in full details. Please feel free to ask "
                "any question and I will always answer it.")

    def build(prompt: str, history: History = (), system: str = "") -> str:
        clean = lambda t: t.strip().replace("\r\n", "\n").replace("\n\n", "\n")
        turns = [("hi", greeting), *[(clean(u), a.strip()) for u, a in history]]
        out = "".join(f"User: {u}{sep}Assistant: {a}{sep}" for u, a in turns)
        return (system + sep if system else "") + out + f"User: {clean(prompt)}{sep}Assistant:"
    return build


_ALPACA = ("Below is an instruction that describes a task. Write a response that appropriately completes the request.\n\n"
           "### Instruction:\n{prompt}\n\n### Response:\n")

# family -> template | builder | CHAT_TEMPLATE.  Names are the reference's example directory names (lower case).
PROMPT_FORMATS: dict[str, Union[str, Callable[..., str]]] = {
    "aquila": "Human: {prompt}###Assistant:",
    "aquila2": "<|startofpiece|>{prompt}<|endofpiece|>",
    "baichuan": "<human>{prompt} <bot>",
    "baichuan2": "<reserved_106> {prompt} <reserved_107>",
    "bluelm": "[|Human|]:{prompt}[|AI|]:",
    "chatglm2": "问：{prompt}\n\n答：",
    "chatglm3": "<|user|>\n{prompt}\n<|assistant|>",
    "chinese-llama2": _llama2,
    "codegeex2": "{prompt}",
    "codegemma": CHAT_TEMPLATE,
    "codellama": "{prompt}\n<FILL_ME>",
    "codeshell": "## human:{prompt}||## assistant:",
    "cohere": "<BOS_TOKEN><|START_OF_TURN_TOKEN|><|USER_TOKEN|>{prompt}<|END_OF_TURN_TOKEN|><|START_OF_TURN_TOKEN|><|CHATBOT_TOKEN|>",
    "decilm-7b": ("### System:\nYou are an AI assistant that follows instruction extremely well. Help as much as you can.\n"
                  "### User:\n{prompt}\n### Assistant:\n"),
    "deepseek": CHAT_TEMPLATE,
    "dolly-v1": _ALPACA,
    "dolly-v2": _ALPACA,
    "falcon": "<human> {prompt} <bot>",
    "flan-t5": "<|User|>:{prompt}",
    "gemma": CHAT_TEMPLATE,
    "gemma2": CHAT_TEMPLATE,
    "glm-edge": CHAT_TEMPLATE,
    "glm4": "<|user|>\n{prompt}\n<|assistant|>",
    "gpt-j": "{prompt}",
    "internlm": "<|User|>:{prompt}\n<|Bot|>:",
    "internlm2": "<|User|>:{prompt}\n<|Bot|>:",
    "llama2": _llama2,
    "llama3": _llama3,
    "llama3.1": _llama3,
    "llama3.2": _llama3,
    "minicpm": CHAT_TEMPLATE,
    "minicpm3": CHAT_TEMPLATE,
    "mistral": "<s>[INST] {prompt} [/INST]",
    "mixtral": "<s>[INST] {prompt} [/INST]",
    "mpt": _chatml(),
    "phi-1_5": " Question:{prompt}\n\n Answer:",
    "phi-2": " Question:{prompt}\n\n Answer:",
    "phi-3": "<|user|>\n{prompt}<|end|>\n<|assistant|>",
    "phixtral": " Question:{prompt}\n\n Answer:",
    "qwen": _chatml("You are a helpful assistant."),
    "qwen1.5": CHAT_TEMPLATE,
    "qwen2": CHAT_TEMPLATE,
    "qwen2.5": CHAT_TEMPLATE,
    "redpajama": "<human>: {prompt}\n<bot>:",
    "replit": "{prompt}",
    "rwkv4": _rwkv("\n\n"),
    "rwkv5": _rwkv("\n"),
    "solar": "<s>### User:\n{prompt}\n### Assistant:\n",
    "stablelm": "<|user|>\n{prompt} <|endoftext|>\n<|assistant|>",
    "starcoder": "{prompt}",
    "vicuna": "### Human:\n{prompt} \n ### Assistant:\n",
    "yi": "{prompt}",
    "yuan2": "{prompt}",
}

# Hugging Face ``config.model_type`` -> family, for callers that have a loaded model instead of a name.  Where several
# families share a model type (llama: llama2 / llama3 / vicuna / yi / solar ...), the checkpoint's own chat template is
# the better source and ``family_of`` returns None unless the vocabulary size tells them apart.
_MODEL_TYPE_FAMILY = {
    "aquila": "aquila", "baichuan": "baichuan2", "bluelm": "bluelm", "chatglm": "chatglm3", "cohere": "cohere",
    "deci": "decilm-7b", "falcon": "falcon", "RefinedWeb": "falcon", "RefinedWebModel": "falcon", "t5": "flan-t5",
    "gemma": "gemma", "gemma2": "gemma2", "glm": "glm-edge", "gptj": "gpt-j", "internlm": "internlm",
    "internlm2": "internlm2", "minicpm": "minicpm", "minicpm3": "minicpm3", "mistral": "mistral", "mixtral": "mixtral",
    "mpt": "mpt", "phi": "phi-2", "phi-msft": "phi-2", "phi3": "phi-3", "phixtral": "phixtral", "qwen": "qwen",
    "qwen2": "qwen2", "gpt_neox": "redpajama", "rwkv": "rwkv4", "rwkv5": "rwkv5", "stablelm": "stablelm",
    "stablelm_epoch": "stablelm", "gpt_bigcode": "starcoder", "yuan": "yuan2", "codeshell": "codeshell",
}


def families() -> Iterable[str]:
    return sorted(PROMPT_FORMATS)


def family_of(config) -> Optional[str]:
    """The family a loaded model's config belongs to, or None when the model type is shared by several families."""
    mt = getattr(config, "model_type", None)
    if mt == "llama":
        vocab = getattr(config, "vocab_size", 0)
        return "llama3" if vocab >= 128000 else None
    return _MODEL_TYPE_FAMILY.get(mt)


def uses_chat_template(family: str) -> bool:
    return PROMPT_FORMATS[_key(family)] is CHAT_TEMPLATE


def _key(family: str) -> str:
    k = family.lower()
    if k not in PROMPT_FORMATS:
        raise KeyError(f"unknown model family {family!r}; known: {', '.join(families())}")
    return k


def build_prompt(family: str, prompt: str, history: History = (), system: str = "", tokenizer=None) -> str:
    """The text a ``family`` checkpoint expects for ``prompt`` (single turn unless ``history`` = [(user, answer), ...]).
    Families whose format lives in the tokenizer need ``tokenizer`` (anything with ``apply_chat_template``)."""
    fmt = PROMPT_FORMATS[_key(family)]
    if fmt is CHAT_TEMPLATE:
        if tokenizer is None or not hasattr(tokenizer, "apply_chat_template"):
            raise ValueError(f"{family}: the prompt format is the tokenizer's chat template; pass tokenizer=")
        messages = [{"role": "system", "content": system}] if system else []
        for user, answer in history:
            messages += [{"role": "user", "content": user}, {"role": "assistant", "content": answer}]
        messages.append({"role": "user", "content": prompt})
        return tokenizer.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
    if callable(fmt):
        return fmt(prompt, history=history, system=system)
    if history or system:
        # single-turn templates: earlier turns are replayed through the same template, the system text goes first
        turns = "".join(fmt.format(prompt=u) + a + "\n" for u, a in history)
        return (system + "\n" if system else "") + turns + fmt.format(prompt=prompt)
    return fmt.format(prompt=prompt)
