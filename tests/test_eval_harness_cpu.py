"""Evaluation / benchmark harness pieces that do not need a device or a dataset download: metrics, prompt builders, the
log-likelihood scorer on a tiny model, the regression gate."""
import csv
import importlib.util
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(rel):
    spec = importlib.util.spec_from_file_location(rel.replace("/", "_")[:-3], os.path.join(ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _CharTok:
    """Character-level tokenizer stub with the call shapes the scripts use."""
    bos_token_id = 1

    def __call__(self, text, add_special_tokens=True, return_tensors=None, truncation=False):
        ids = ([1] if add_special_tokens else []) + [3 + (ord(c) % 300) for c in text]

        class R:
            input_ids = torch.tensor([ids]) if return_tensors == "pt" else ids
        return R

    def decode(self, ids, skip_special_tokens=True):
        return "".join(chr(97 + int(i) % 26) for i in ids)


@pytest.fixture(scope="module")
def tiny_model():
    from transformers import LlamaConfig, LlamaForCausalLM
    from bigdl_amd.optimize import optimize_model
    torch.manual_seed(0)
    os.environ["IPEX_LLM_LAST_LM_HEAD"] = "0"
    cfg = LlamaConfig(vocab_size=320, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=512)
    m = optimize_model(LlamaForCausalLM(cfg).eval(), low_bit="sym_int8")
    yield m
    os.environ.pop("IPEX_LLM_LAST_LM_HEAD", None)


def test_loglikelihood_scorer_matches_manual_sum(tiny_model):
    ev = _load("benchmark/harness/evaluate.py")
    tok = _CharTok()
    ll = ev.loglikelihood(tiny_model, tok, "the sky is", " blue")
    ids = tok("the sky is").input_ids + tok(" blue", add_special_tokens=False).input_ids
    with torch.no_grad():
        lp = torch.log_softmax(tiny_model(torch.tensor([ids])).logits[0].float(), -1)
    want = sum(float(lp[i - 1, ids[i]]) for i in range(len(ids) - 5, len(ids)))
    assert abs(ll - want) < 1e-3
    rows = [{"context": "q: 1+1=", "choices": [" 2", " 3", " 22"], "gold": 0}] * 3
    res = ev.score_task(tiny_model, tok, rows)
    assert res["n"] == 3 and 0.0 <= res["acc"] <= 1.0 and 0.0 <= res["acc_norm"] <= 1.0


def test_ceval_prompt_and_accuracy(tmp_path, tiny_model):
    ce = _load("benchmark/ceval/eval.py")
    for split, n in (("dev", 2), ("val", 4)):
        os.makedirs(tmp_path / split)
        with open(tmp_path / split / f"physics_{split}.csv", "w", newline="", encoding="utf-8") as f:
            w = csv.writer(f)
            w.writerow(["id", "question", "A", "B", "C", "D", "answer"])
            for i in range(n):
                w.writerow([i, f"q{i}?", "a", "b", "c", "d", "ABCD"[i % 4]])
    rows = ce.read_csv(str(tmp_path / "val" / "physics_val.csv"))
    p = ce.build_prompt("physics", ce.read_csv(str(tmp_path / "dev" / "physics_dev.csv")), rows[0], k=2)
    assert p.count("答案：") == 3 and p.endswith("答案：") and "答案：A\n\n" in p
    res = ce.evaluate(tiny_model, _CharTok(), str(tmp_path), "val", k=2)
    assert res["physics"]["n"] == 4 and 0.0 <= res["_average"]["acc"] <= 1.0


def test_wer_and_longbench_metrics():
    wh = _load("benchmark/whisper/run_whisper.py")
    assert wh.wer(["the cat sat on the mat"], ["the cat sat on the mat"]) == 0.0
    assert abs(wh.wer(["the cat sat on the mat"], ["the cat sit on mat"]) - 2 / 6) < 1e-9          # 1 sub + 1 del
    assert wh.wer(["Hello, World!"], ["hello world"]) == 0.0
    lb = _load("benchmark/longbench/pred.py")
    assert lb.f1_score("The Eiffel Tower", "eiffel tower") == 1.0
    assert 0.0 < lb.f1_score("paris france", "paris") < 1.0
    assert abs(lb.rouge_l("a b c d", "a c d e") - 0.75) < 1e-9
    assert lb.classification_score("this is a location question", "location", ["location", "number"]) == 1.0
    assert lb.count_score("there are 3 unique paragraphs, maybe 4", "3") == 0.5
    assert lb.score("trec", "number\nsecond line location", ["number"], ["location", "number"]) == 1.0

    class Tok(_CharTok):
        def decode(self, ids, skip_special_tokens=True):
            return "x" * len(ids)
    assert len(lb.truncate_middle(Tok(), "a" * 100, 20)) == 20


def test_regression_gate(tmp_path):
    rg = _load("benchmark/regression/check.py")
    cols = ["model", "1st token avg latency (ms)", "2+ avg latency (ms/token)", "input/output tokens", "batch_size", "low_bit",
            "test_api"]

    def write(name, rows):
        with open(tmp_path / name, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(cols)
            w.writerows(rows)
        return str(tmp_path / name)
    base = write("b.csv", [["m", 10.0, 1.0, "32-32", 1, "sym_int4", "api"], ["m", 20.0, 2.0, "1024-128", 1, "sym_int4", "api"]])
    good = write("g.csv", [["m", 10.2, 1.04, "32-32", 1, "sym_int4", "api"], ["m", 19.0, 1.9, "1024-128", 1, "sym_int4", "api"]])
    slow = write("s.csv", [["m", 10.0, 1.08, "32-32", 1, "sym_int4", "api"], ["m", 20.0, 2.0, "1024-128", 1, "sym_int4", "api"]])
    part = write("p.csv", [["m", 10.0, 1.0, "32-32", 1, "sym_int4", "api"]])
    assert rg.main(["-b", base, "-c", good, "--html", str(tmp_path / "o.html")]) == 0
    assert "<table" in open(tmp_path / "o.html").read()
    assert rg.main(["-b", base, "-c", slow]) == 1          # +8 % next-token latency
    assert rg.main(["-b", base, "-c", part]) == 1          # a row is missing
    st = _load("benchmark/stress/run_stress_test.py")
    a, b, d = st.drift([1.0, 1.0, 1.0, 1.0, 1.1, 1.1, 1.1, 1.1])
    assert abs(d - 0.1) < 1e-9
