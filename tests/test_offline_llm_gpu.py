"""The vLLM-style offline interface on the GPU: prompts of different lengths, more of them than cache slots, go through
the continuous batcher and come back in order with the tokens the engine generates for each prompt alone."""
import pytest
import torch
from transformers import LlamaConfig, LlamaForCausalLM

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny_dir(tmp_path_factory):
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                      num_attention_heads=4, num_key_value_heads=2, head_dim=64, max_position_embeddings=512,
                      tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg).eval()
    m.generation_config.eos_token_id = None
    d = str(tmp_path_factory.mktemp("tiny_llama_offline"))
    m.save_pretrained(d)
    return d


def test_offline_generate_runs_on_the_batcher_and_matches_single_prompt_generation(tiny_dir):
    from bigdl_amd.models.hf_patch import get_native_engine
    from bigdl_amd.transformers import AutoModelForCausalLM
    from bigdl_amd.vllm import LLM, SamplingParams
    model = AutoModelForCausalLM.from_pretrained(tiny_dir, load_in_low_bit="sym_int4")
    llm = LLM(model=(model, None), max_num_seqs=3, max_model_len=128, enforce_eager=True, gpu_memory_utilization=0.5)
    assert llm.batcher is not None and llm.device == "cuda"
    g = torch.Generator().manual_seed(3)
    lens = [5, 17, 9, 33, 12, 21, 7]
    prompts = [torch.randint(3, 320, (n,), generator=g).tolist() for n in lens]
    budgets = [6, 9, 4, 7, 12, 5, 8]
    sps = [SamplingParams(temperature=0.0, max_tokens=b, ignore_eos=True) for b in budgets]
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=sps)
    assert [o.prompt_token_ids for o in outs] == prompts
    assert llm.batcher.steps > max(budgets)                    # 7 requests through 3 slots: several rounds of admission
    # (the reference runs on an engine of its own: the batcher owns the slot-mode cache of the served engine)
    eng = get_native_engine(AutoModelForCausalLM.from_pretrained(tiny_dir, load_in_low_bit="sym_int4").half().to("cuda"))
    agree = total = 0
    for o, p, b in zip(outs, prompts, budgets):
        got = o.outputs[0].token_ids
        assert len(got) == b and o.outputs[0].finish_reason == "length"
        want = eng.generate(torch.tensor([p]), max_new_tokens=b)[0, len(p):].cpu().tolist()
        assert got[0] == want[0]
        agree += sum(x == y for x, y in zip(got, want))
        total += b
    # batched rows take the multi-row kernels: same tokens up to rare fp16 ties
    assert agree / total >= 0.9, (agree, total)
    # a stop token id ends a request early and is reported as "stop"
    first = outs[1].outputs[0].token_ids
    cut = llm.generate(prompt_token_ids=[prompts[1]], sampling_params=SamplingParams(temperature=0.0, max_tokens=9, ignore_eos=True,
                                                                                       stop_token_ids=[first[2]]))[0].outputs[0]
    assert cut.finish_reason == "stop" and cut.token_ids[-1] == first[2] and len(cut.token_ids) <= 3
    # sampling runs and respects the budget
    s = llm.generate(prompt_token_ids=prompts[:2], sampling_params=SamplingParams(temperature=0.8, top_p=0.9, top_k=20, max_tokens=5,
                                                                                 ignore_eos=True, seed=1))
    assert all(len(o.outputs[0].token_ids) == 5 for o in s)


def test_reference_style_script_with_the_xpu_device_name(tiny_dir):
    """What the reference's GPU examples do verbatim -- ``.to('xpu')``, ``torch.xpu.synchronize()`` -- after
    ``enable_xpu_alias()``: the model lands on the GPU of this build and generates the same tokens as with 'cuda'."""
    import ipex_llm
    from ipex_llm.transformers import AutoModelForCausalLM
    ids = torch.randint(3, 320, (1, 12))
    model = AutoModelForCausalLM.from_pretrained(tiny_dir, load_in_4bit=True).half().to("cuda")
    want = model.generate(ids.to("cuda"), max_new_tokens=8, do_sample=False).cpu()
    del model
    assert ipex_llm.enable_xpu_alias() == "cuda"
    try:
        model = AutoModelForCausalLM.from_pretrained(tiny_dir, load_in_4bit=True).half().to("xpu")
        assert next(model.parameters()).device.type == "cuda"
        got = model.generate(ids.to("xpu"), max_new_tokens=8, do_sample=False)
        torch.xpu.synchronize()
        assert torch.xpu.max_memory_allocated() > 0 and torch.zeros(1, device="xpu:0").is_cuda
        assert torch.equal(got.cpu(), want)
    finally:
        ipex_llm.disable_xpu_alias()
