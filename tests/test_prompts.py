"""Prompt-format table (bigdl_amd/utils/prompts.py) against the strings the reference's per-family example scripts build
(python/llm/example/GPU/HuggingFace/LLM/<family>/generate.py)."""
import os

import pytest

from bigdl_amd.utils import prompts as P

REF = "/root/reference/python/llm/example/GPU/HuggingFace/LLM"


def test_every_reference_family_has_a_format():
    want = {"aquila", "aquila2", "baichuan", "baichuan2", "bluelm", "chatglm2", "chatglm3", "chinese-llama2", "codegeex2",
            "codegemma", "codellama", "codeshell", "cohere", "decilm-7b", "deepseek", "dolly-v1", "dolly-v2", "falcon", "flan-t5",
            "gemma", "gemma2", "glm-edge", "glm4", "gpt-j", "internlm", "internlm2", "llama2", "llama3", "llama3.1", "llama3.2",
            "minicpm", "minicpm3", "mistral", "mixtral", "mpt", "phi-1_5", "phi-2", "phi-3", "phixtral", "qwen", "qwen1.5",
            "qwen2", "qwen2.5", "redpajama", "replit", "rwkv4", "rwkv5", "solar", "stablelm", "starcoder", "vicuna", "yi", "yuan2"}
    assert want <= set(P.families())
    if os.path.isdir(REF):
        dirs = {d.lower() for d in os.listdir(REF) if os.path.isdir(os.path.join(REF, d))}
        assert dirs <= set(P.families()), dirs - set(P.families())


def test_single_turn_templates():
    assert P.build_prompt("aquila", "hi") == "Human: hi###Assistant:"
    assert P.build_prompt("chatglm3", "hi") == "<|user|>\nhi\n<|assistant|>"
    assert P.build_prompt("Mistral", "hi") == "<s>[INST] hi [/INST]"
    assert P.build_prompt("phi-3", "hi") == "<|user|>\nhi<|end|>\n<|assistant|>"
    assert P.build_prompt("starcoder", "def f():") == "def f():"
    assert P.build_prompt("dolly-v2", "hi").endswith("### Instruction:\nhi\n\n### Response:\n")


def test_llama2_and_llama3_builders():
    assert P.build_prompt("llama2", "What is AI?") == "<s>[INST] <<SYS>>\n\n<</SYS>>\n\nWhat is AI? [/INST]"
    two = P.build_prompt("llama2", " more ", history=[("first", " answer ")], system="S")
    assert two == "<s>[INST] <<SYS>>\nS\n<</SYS>>\n\nfirst [/INST] answer </s><s>[INST] more [/INST]"
    l3 = P.build_prompt("llama3.1", "q", system="S", history=[("a", "b")])
    assert l3 == ("<|begin_of_text|><|start_header_id|>system<|end_header_id|>\n\nS<|eot_id|>"
                  "<|start_header_id|>user<|end_header_id|>\n\na<|eot_id|><|start_header_id|>assistant<|end_header_id|>\n\nb<|eot_id|>"
                  "<|start_header_id|>user<|end_header_id|>\n\nq<|eot_id|><|start_header_id|>assistant<|end_header_id|>\n\n")
    assert P.build_prompt("llama3", "q").startswith("<|begin_of_text|><|start_header_id|>user")


def test_chatml_and_rwkv():
    assert P.build_prompt("mpt", "hi") == "<|im_start|>user\nhi<|im_end|>\n<|im_start|>assistant\n"
    q = P.build_prompt("qwen", "hi")
    assert q.startswith("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n") and q.endswith("<|im_start|>assistant\n")
    r4 = P.build_prompt("rwkv4", "a\n\nb")
    assert r4.startswith("User: hi\n\nAssistant: Hi. I am your assistant") and r4.endswith("\n\nUser: a\nb\n\nAssistant:")
    assert P.build_prompt("rwkv5", "x").endswith("answer it.\nUser: x\nAssistant:")


def test_chat_template_families_need_the_tokenizer():
    assert P.uses_chat_template("qwen2") and not P.uses_chat_template("llama2")
    with pytest.raises(ValueError):
        P.build_prompt("gemma", "hi")

    class Tok:
        def apply_chat_template(self, messages, tokenize, add_generation_prompt):
            assert not tokenize and add_generation_prompt
            return "|".join(f"{m['role']}:{m['content']}" for m in messages) + "|assistant:"
    assert P.build_prompt("qwen2.5", "q", system="s", history=[("a", "b")], tokenizer=Tok()) == "system:s|user:a|assistant:b|user:q|assistant:"
    with pytest.raises(KeyError):
        P.build_prompt("no-such-family", "hi")


def test_family_of_config():
    class C:
        def __init__(self, **k):
            self.__dict__.update(k)
    assert P.family_of(C(model_type="mistral")) == "mistral"
    assert P.family_of(C(model_type="llama", vocab_size=128256)) == "llama3"
    assert P.family_of(C(model_type="llama", vocab_size=32000)) is None
    assert P.family_of(C(model_type="unknown")) is None
