"""The vLLM-style offline interface (bigdl_amd/vllm/offline.py) on the CPU: the module-path fallback, the class that
``IPEXLLMClass`` resolves to without vLLM, and the LangChain wrapper over it.  The batched engine path is covered by
tests/test_offline_llm_gpu.py."""
import importlib.util

import pytest
import torch
from transformers import LlamaConfig, LlamaForCausalLM

from bigdl_amd.vllm import LLM, SamplingParams


class CharTokenizer:
    """One token per character (ids 3 ..), enough of the tokenizer protocol for the wrappers."""
    eos_token_id = 2

    def __call__(self, text, **kw):
        class Enc:
            input_ids = [3 + (ord(c) % 120) for c in text]
        return Enc()

    def decode(self, ids, skip_special_tokens=True):
        return "".join(chr(97 + (int(t) - 3) % 26) for t in ids if not (skip_special_tokens and int(t) == self.eos_token_id))


def _tiny():
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=128, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=128)
    m = LlamaForCausalLM(cfg).eval()
    m.generation_config.eos_token_id = None
    return m


def test_module_path_generation_matches_hf_generate():
    model = _tiny()
    llm = LLM(model=(model, CharTokenizer()), device="cpu", max_model_len=64)
    prompts = [[5, 9, 33, 7], {"prompt_token_ids": [11, 12, 13, 14, 15, 16]}, "hello"]
    outs = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True, presence_penalty=0.0))
    assert [o.prompt for o in outs] == [None, None, "hello"]
    for o in outs:
        want = model.generate(torch.tensor([o.prompt_token_ids]), max_new_tokens=6, do_sample=False,
                              eos_token_id=None)[0, len(o.prompt_token_ids):].tolist()
        assert o.outputs[0].token_ids == want
        assert o.outputs[0].finish_reason == "length" and o.finished
        assert o.outputs[0].text == CharTokenizer().decode(want)


def test_stop_strings_token_budget_and_per_prompt_params():
    model = _tiny()
    tok = CharTokenizer()
    llm = LLM(model=(model, tok), device="cpu", max_model_len=32)
    full = llm.generate([[5, 6, 7]], SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))[0].outputs[0]
    stop = full.text[3:5]
    a, b = llm.generate([[5, 6, 7], [5, 6, 7]], [SamplingParams(temperature=0, max_tokens=8, ignore_eos=True, stop=stop),
                                               SamplingParams(temperature=0, max_tokens=None, ignore_eos=True)])
    assert a.outputs[0].text == full.text[:full.text.find(stop)] and a.outputs[0].finish_reason == "stop"
    assert len(b.outputs[0].token_ids) == 32 - 3              # no budget given: up to max_model_len
    with pytest.raises(ValueError):
        SamplingParams(n=2)


@pytest.mark.skipif(importlib.util.find_spec("vllm") is not None, reason="with vLLM installed the class is vLLM's LLM")
def test_engine_entry_point_resolves_to_the_offline_class_without_vllm():
    from bigdl_amd.vllm.engine import __getattr__ as engine_attr
    from ipex_llm.vllm.xpu.engine import IPEXLLMClass
    assert engine_attr("IPEXLLMClass") is LLM and IPEXLLMClass is LLM


def test_langchain_wrapper_batches_all_prompts_into_one_generation():
    from bigdl_amd.langchain.vllm import VLLM
    model = _tiny()
    client = LLM(model=(model, CharTokenizer()), device="cpu", max_model_len=64)
    calls = []
    orig = client.generate
    client.generate = lambda prompts, sp, **kw: (calls.append(len(prompts)), orig(prompts, sp, **kw))[1]
    llm = VLLM(client=client, max_new_tokens=5, temperature=0.0, ignore_eos=True)
    texts = llm.batch(["abc", "hello there", "xy"])
    assert calls == [3] and all(len(t) == 5 for t in texts)
    assert llm.invoke("abc") == texts[0]
    stop = texts[0][2:4]
    assert llm.invoke("abc", stop=[stop]) == texts[0][:texts[0].find(stop)]


@pytest.mark.skipif(importlib.util.find_spec("vllm") is not None, reason="with vLLM installed vLLM's own server runs")
def test_openai_server_entry_point_starts_the_builtin_batched_server(monkeypatch):
    """vLLM's command line on the built-in server: the options it understands keep their meaning, the rest is reported."""
    import bigdl_amd.serving.fastapi.api_server as builtin
    from ipex_llm.vllm.cpu.entrypoints.openai.api_server import main as cpu_main
    from ipex_llm.vllm.xpu.entrypoints.openai.api_server import main
    seen = []
    monkeypatch.setattr(builtin, "serve", lambda *a, **kw: seen.append((a, kw)))
    main(["--model", "/m", "--served-model-name", "x", "--port", "8123", "--load-in-low-bit", "fp8", "--max-num-seqs", "12",
          "--max-model-len", "2048", "--gpu-memory-utilization", "0.9", "--enforce-eager"])
    cpu_main(["--model", "/m"])
    assert seen[0] == (("/m", "fp8", 8123, "0.0.0.0", "x"), {"max_num_seqs": 12, "max_seq_len": 2048})
    assert seen[1] == (("/m", "sym_int4", 8000, "0.0.0.0", None), {"max_num_seqs": 64, "max_seq_len": 4096})
    with pytest.raises(SystemExit):
        main(["--model", "/m", "--tensor-parallel-size", "2"])
