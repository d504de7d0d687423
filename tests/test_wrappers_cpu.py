"""LangChain / LlamaIndex / FastChat wrappers on a tiny low-bit model (the third-party base packages are absent on
the test boxes; the wrappers then run on their local stand-in bases with the same call surface)."""
import json

import pytest
import torch
from tokenizers import Tokenizer
from tokenizers.models import WordLevel
from tokenizers.pre_tokenizers import Whitespace
from transformers import LlamaConfig, LlamaForCausalLM, PreTrainedTokenizerFast


@pytest.fixture(scope="module")
def tiny(tmp_path_factory):
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=256, tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg).eval()
    m.generation_config.eos_token_id = None
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    vocab.update({f"w{i}": i + 3 for i in range(317)})
    t = Tokenizer(WordLevel(vocab, unk_token="<unk>"))
    t.pre_tokenizer = Whitespace()
    tok = PreTrainedTokenizerFast(tokenizer_object=t, bos_token="<s>", eos_token="</s>", unk_token="<unk>")
    d = str(tmp_path_factory.mktemp("tiny_wrap"))
    m.save_pretrained(d)
    tok.save_pretrained(d)
    return d


def _ref(d, prompt, n):
    from transformers import AutoTokenizer
    from bigdl_amd.transformers import AutoModelForCausalLM
    tok = AutoTokenizer.from_pretrained(d)
    m = AutoModelForCausalLM.from_pretrained(d, load_in_4bit=True)
    ids = tok(prompt, return_tensors="pt").input_ids
    out = m.generate(ids, max_new_tokens=n, do_sample=False)
    return tok.decode(out[0, ids.shape[1]:], skip_special_tokens=True)


def test_langchain_llm_and_embeddings(tiny):
    from ipex_llm.langchain.embeddings import TransformersEmbeddings
    from ipex_llm.langchain.llms import TransformersLLM
    llm = TransformersLLM.from_model_id(tiny, model_kwargs={"trust_remote_code": True})
    assert llm.invoke("w1 w2 w3", max_new_tokens=6, do_sample=False) == _ref(tiny, "w1 w2 w3", 6)
    full = llm.invoke("w1 w2 w3", max_new_tokens=6, do_sample=False)
    stop = full.split()[2]
    assert stop not in llm.invoke("w1 w2 w3", stop=[stop], max_new_tokens=6, do_sample=False)
    emb = TransformersEmbeddings.from_model_id(tiny)
    v = emb.embed_query("w4 w5")
    assert len(v) == 128 and len(emb.embed_documents(["w1", "w2 w3"])) == 2


def test_llamaindex_llm(tiny):
    from ipex_llm.llamaindex.llms import IpexLLM
    llm = IpexLLM.from_model_id(tiny, max_new_tokens=5, generate_kwargs={"do_sample": False})
    assert llm.complete("w7 w8").text == _ref(tiny, "w7 w8", 5)
    pieces = list(llm.stream_complete("w7 w8"))
    assert pieces[-1].text == llm.complete("w7 w8").text
    assert llm.metadata.num_output == 5


def test_fastchat_worker_protocol(tiny):
    from ipex_llm.serving.fastchat.ipex_llm_worker import BigDLLLMWorker
    w = BigDLLLMWorker(model_path=tiny, load_in_low_bit="sym_int4", device="cpu", stream_interval=2)
    chunks = list(w.generate_stream_gate({"prompt": "w1 w2", "max_new_tokens": 6, "temperature": 0.0, "echo": False}))
    assert all(c.endswith(b"\0") for c in chunks)
    last = json.loads(chunks[-1][:-1])
    assert last["text"] == _ref(tiny, "w1 w2", 6) and last["usage"]["completion_tokens"] == 6
    assert last["finish_reason"] == "length" and last["error_code"] == 0
    assert w.generate_gate({"prompt": "w1 w2", "max_new_tokens": 6, "temperature": 0.0, "echo": False})["text"] == last["text"]
    e = w.get_embeddings({"input": ["w1 w2", "w3"]})
    assert len(e["embedding"]) == 2 and e["token_num"] == 3


def test_fastchat_batched_worker_protocol(tiny):
    """The batching FastChat worker (the reference's vllm_worker.py role): NUL-terminated JSON chunks, stop strings
    that abort the request, two requests in flight on one engine loop, the worker endpoints."""
    import asyncio
    from fastapi.testclient import TestClient
    from ipex_llm.serving.fastchat.vllm_worker import VLLMWorker, create_app
    w = VLLMWorker(model_path=tiny, load_in_low_bit="sym_int4")
    want = _ref(tiny, "w1 w2", 6)

    async def collect(params):
        return [json.loads(c[:-1]) async for c in w.generate_stream(params)]

    async def both():
        a = collect({"prompt": "w1 w2", "max_new_tokens": 6, "temperature": 0.0, "echo": False})
        b = collect({"prompt": "w1 w2", "max_new_tokens": 6, "temperature": 0.0, "echo": True, "stop": want.split()[3]})
        res = await asyncio.gather(a, b)
        w._driver.cancel()
        return res
    plain, stopped = asyncio.run(both())
    assert plain[-1]["text"] == want and plain[-1]["finish_reason"] == "length" and plain[-1]["usage"]["completion_tokens"] == 6
    assert all(c["finish_reason"] is None for c in plain[:-1])
    assert stopped[-1]["finish_reason"] == "stop" and stopped[-1]["text"] == "w1 w2" + " ".join(want.split()[:3]) + " "
    w._driver = None
    with TestClient(create_app(w)) as client:
        assert client.post("/count_token", json={"prompt": "w1 w2 w3"}).json()["count"] == 3
        assert client.post("/model_details").json()["context_length"] == 256
        assert client.post("/worker_get_status").json()["queue_length"] == 0
        out = client.post("/worker_generate", json={"prompt": "w1 w2", "max_new_tokens": 6, "temperature": 0.0, "echo": False}).json()
        assert out["text"] == want


def test_vllm_hook_requires_vllm():
    from ipex_llm.vllm.model_convert import _ipex_llm_convert
    with pytest.raises(ImportError):
        _ipex_llm_convert("sym_int4")


def test_fastchat_model_adapters_and_patched_loaders(tiny, tmp_path, monkeypatch):
    """The adapters FastChat is given (reference serving/fastchat/bigdl_llm_model.py): path patterns, loading through
    this package, the low-bit directory adapter, and what ``patch_fastchat`` rebinds (against a stand-in of
    ``fastchat.model.model_adapter``: FastChat is not installed here, parity with it unpinned beyond this protocol)."""
    import shutil
    import sys
    import types
    from ipex_llm.serving.fastchat import bigdl_llm_model as M
    from bigdl_amd.transformers import AutoModelForCausalLM
    from bigdl_amd.transformers.low_bit_linear import LowBitLinear
    low, conv = M.BigDLLMLOWBITAdapter(), M.BigDLLLMAdapter()
    assert conv.match("/models/bigdl-7b") and not conv.match("/models/llama")
    assert low.match("/x/bigdl-lowbit-7b") and not low.match("/x/bigdl-7b")
    d = str(tmp_path / "bigdl-tiny")
    shutil.copytree(tiny, d)
    model, tok = conv.load_model(d, {"revision": "main"})
    assert isinstance(model.model.layers[0].mlp.down_proj, LowBitLinear) and tok("w1 w2").input_ids
    ld = str(tmp_path / "bigdl-lowbit-tiny")
    model.save_low_bit(ld)
    tok.save_pretrained(ld)
    again, _ = low.load_model(ld, {})
    ids = torch.tensor([[5, 6, 7, 8]])
    with torch.no_grad():
        torch.testing.assert_close(again(ids).logits, model(ids).logits)
    # the module-level loader picks the adapter by pattern and falls back to "any causal LM, converted on load"
    for path in (d, ld, tiny):
        m, _ = M.load_model(path, device="cpu")
        assert isinstance(m.model.layers[0].mlp.down_proj, LowBitLinear)
    # patch_fastchat against a stand-in of FastChat's adapter registry
    fake = types.ModuleType("fastchat.model.model_adapter")

    class ChatGLMAdapter(M.BaseModelAdapter):
        def load_model(self, *a):
            return "orig-glm"
    fake.ChatGLMAdapter, fake.BaseModelAdapter = ChatGLMAdapter, M.BaseModelAdapter
    fake.model_adapters = [M.BaseModelAdapter()]
    fake.register_model_adapter = lambda cls: fake.model_adapters.append(cls())
    fake.load_model = lambda *a, **k: "orig-load"
    pkg, sub = types.ModuleType("fastchat"), types.ModuleType("fastchat.model")
    pkg.model, sub.model_adapter = sub, fake
    for name, mod in (("fastchat", pkg), ("fastchat.model", sub), ("fastchat.model.model_adapter", fake)):
        monkeypatch.setitem(sys.modules, name, mod)
    M.patch_fastchat()
    try:
        assert [type(a).__name__ for a in fake.model_adapters[:2]] == ["BigDLLMLOWBITAdapter", "BigDLLLMAdapter"]
        assert fake.load_model is M.load_model and ChatGLMAdapter.load_model is M.load_model_chatglm
        assert M.BaseModelAdapter.load_model is M.load_model_base
        M.patch_fastchat()                                       # idempotent
        assert len(fake.model_adapters) == 3
    finally:
        M.unpatch_fastchat()
    assert fake.load_model() == "orig-load" and ChatGLMAdapter().load_model() == "orig-glm"
    assert not M.is_fastchat_patched
