"""fp16 decode, self-speculative decode and prompt-lookup decode on the native engine at a named model shape:
the ``bigdl_fp16_gpu`` / ``speculative_gpu`` / ``lookup_gpu`` modes of the reference's all-in-one benchmark
(python/llm/dev/benchmark/all-in-one/run.py:1101-1170, 1466-1560, 1830-1950), one JSON line per mode.

    python benchmark/spec_bench.py --model llama2-7b --in-len 1024 --out-len 128 [--draft sym_int4] [--modes fp16,spec,lookup]

Weights are random-init (no checkpoints on the box): the fp16 target is drawn first, the draft is the host
quantizer's low-bit conversion of those same weights -- the relationship a real self-speculative pair has.  What a
random network cannot supply is the peaked next-token distribution of a trained one: its top-1 margin is tiny, so
quantization noise flips the argmax far more often than on a trained model.  The acceptance rate printed here is
therefore a LOWER bound for trained weights; the ``speculative_gpu_if_all_accepted`` line prices a round of k drafts +
one verification at its measured time and credits it with k + 1 tokens, which bounds the speed-up from above.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama2-7b")
    ap.add_argument("--in-len", type=int, default=1024)
    ap.add_argument("--out-len", type=int, default=128)
    ap.add_argument("--draft", default="sym_int4")
    ap.add_argument("--max-step-draft", type=int, default=6)
    ap.add_argument("--modes", default="fp16,int4,spec,lookup")
    ap.add_argument("--trials", type=int, default=3)
    ap.add_argument("--oracle-rates", default="0.3,0.5,0.7,0.8,0.9,1.0", help="per-token acceptance rates of the oracle mode")
    ap.add_argument("--layers", type=int, default=0, help="override the layer count (quick runs)")
    ap.add_argument("--draft-layers", default="", help="oracle_skip mode: comma-separated numbers of layers the layer-skipping "
                                                        "low-bit draft keeps (e.g. 8,12,16)")
    a = ap.parse_args()
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models import decoder as D
    from bigdl_amd.models.spec_engine import SelfSpeculativeEngine, quantized_draft_of
    table = {"llama2-7b": D.LLAMA2_7B, "tinyllama": D.TINY_LLAMA, "llama3-8b": D.LLAMA3_8B}
    d = dict(table[a.model])
    if a.layers:
        d["num_layers"] = a.layers
    cfg = D.DecoderConfig(**d)
    modes = a.modes.split(",")
    t0 = time.perf_counter()
    target = D.build_random_decoder(cfg, Q.FP16, device="cuda", seed=1234)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(3, cfg.vocab_size, (1, a.in_len), generator=g).cuda()

    def emit(mode, first, rest, **extra):
        print(json.dumps({"mode": mode, "model": a.model, "in_out": f"{a.in_len}-{a.out_len}",
                          "first_token_ms": round(first * 1e3, 3), "next_token_ms": round(rest * 1e3, 4), **extra}), flush=True)

    base_rest = None
    if "fp16" in modes:
        for _ in range(a.trials + 1):
            ref = target.generate(ids, max_new_tokens=a.out_len)
        base_rest = target.rest_cost_mean
        emit("bigdl_fp16_gpu", target.first_cost, target.rest_cost_mean, peak_mem_gb=round(target.peak_memory, 2),
             weight_gb=round(target.weight_bytes() / 2 ** 30, 2), build_s=round(t_build, 1))
    draft = None
    if "spec" in modes or "int4" in modes or "oracle" in modes:
        t0 = time.perf_counter()
        draft = quantized_draft_of(target, Q.ggml_tensor_qtype[a.draft])
        t_quant = time.perf_counter() - t0
    if "int4" in modes:
        for _ in range(a.trials + 1):
            draft.generate(ids, max_new_tokens=a.out_len)
        emit(f"draft_alone_{a.draft}", draft.first_cost, draft.rest_cost_mean, quantize_s=round(t_quant, 1))
        draft.cache = None
        draft._graphs.clear()
        draft._static.clear()
    if "spec" in modes:
        spec = SelfSpeculativeEngine(target, draft, max_step_draft=a.max_step_draft)
        for _ in range(a.trials + 1):
            out = spec.generate(ids, max_new_tokens=a.out_len)
        same = bool(torch.equal(out, ref)) if "fp16" in modes else None
        emit("speculative_gpu", spec.first_cost, spec.rest_cost_mean, draft=a.draft, n_drafted=spec.n_drafted,
             n_matched=spec.n_matched, accept_rate=round(spec.n_matched / max(1, spec.n_drafted), 3),
             rounds=len(spec.accept_num), tokens_per_round=round(sum(spec.accept_num) / max(1, len(spec.accept_num)), 2),
             ms_per_round=round(1e3 * sum(spec.generate_time) / max(1, len(spec.generate_time)), 3),
             equals_plain_fp16=same, speedup_vs_fp16=None if base_rest is None else round(base_rest / spec.rest_cost_mean, 3))
        # upper bound: measured time of the rounds that drafted k tokens, credited with k + 1 tokens
        k = a.max_step_draft
        per_round = [t for t, n in zip(spec.generate_time, spec.draft_num) if n == k]
        if per_round:
            full = sum(per_round) / len(per_round)
            emit("speculative_gpu_if_all_accepted", spec.first_cost, full / (k + 1), k=k,
                 ms_per_round=round(full * 1e3, 3),
                 speedup_vs_fp16=None if base_rest is None else round(base_rest / (full / (k + 1)), 3))
    if "oracle" in modes:
        # controlled acceptance: every draft token is accepted with probability a, whatever the models say (the text is
        # meaningless, the time per generated token is what a real pair with that per-token acceptance rate gets)
        if draft is None:
            draft = quantized_draft_of(target, Q.ggml_tensor_qtype[a.draft])
        spec = SelfSpeculativeEngine(target, draft, max_step_draft=a.max_step_draft)
        for acc in [float(x) for x in a.oracle_rates.split(",")]:
            best = None
            for _ in range(a.trials + 1):
                spec.generate(ids, max_new_tokens=a.out_len, oracle_accept=acc)
                cur = (spec.rest_cost_mean, spec.n_matched / max(1, spec.n_drafted),
                       sum(spec.accept_num) / max(1, len(spec.accept_num)))
                best = cur if best is None or cur[0] < best[0] else best
            emit("speculative_gpu_controlled_acceptance", spec.first_cost, best[0], draft=a.draft, per_token_acceptance=acc,
                 measured_accept_rate=round(best[1], 3), tokens_per_round=round(best[2], 2), k_max=a.max_step_draft,
                 speedup_vs_fp16=None if base_rest is None else round(base_rest / best[0], 3))
    if "oracle_skip" in modes and a.draft_layers:
        # the same table for low-bit drafts that also skip layers (layer_skip_draft_of): cheaper drafts, and the
        # acceptance a real model gives such a draft is a property of the model and of the layers skipped
        from bigdl_amd.models.spec_engine import default_kept_layers, layer_skip_draft_of
        if draft is None:
            draft = quantized_draft_of(target, Q.ggml_tensor_qtype[a.draft])
        for keep in [int(x) for x in a.draft_layers.split(",")]:
            kept = default_kept_layers(cfg.num_layers, keep)
            skip = layer_skip_draft_of(draft, kept)
            spec = SelfSpeculativeEngine(target, skip, max_step_draft=a.max_step_draft)
            plain = spec.generate(ids, max_new_tokens=a.out_len)          # real comparison: must equal plain greedy
            same = bool(torch.equal(plain, ref)) if "fp16" in modes else None
            real = (spec.rest_cost_mean, spec.n_matched / max(1, spec.n_drafted))
            for acc in [float(x) for x in a.oracle_rates.split(",")]:
                best = None
                for _ in range(a.trials + 1):
                    spec.generate(ids, max_new_tokens=a.out_len, oracle_accept=acc)
                    cur = (spec.rest_cost_mean, spec.n_matched / max(1, spec.n_drafted),
                           sum(spec.accept_num) / max(1, len(spec.accept_num)),
                           sum(spec.draft_time) / max(1, sum(spec.draft_num)))
                    best = cur if best is None or cur[0] < best[0] else best
                emit("speculative_gpu_layer_skip_controlled_acceptance", spec.first_cost, best[0], draft=a.draft,
                     draft_layers=keep, kept_layers=kept, per_token_acceptance=acc, measured_accept_rate=round(best[1], 3),
                     tokens_per_round=round(best[2], 2), k_max=a.max_step_draft, equals_plain_fp16=same,
                     random_init_accept_rate=round(real[1], 3),
                     speedup_vs_fp16=None if base_rest is None else round(base_rest / best[0], 3))
    if "lookup" in modes:
        from bigdl_amd.transformers.lookup import lookup_generate
        from bigdl_amd.transformers.stepper import NativeStepper

        class Holder:                      # lookup_generate keeps its statistics on the object it is bound to
            generation_config = None
        for name, prompt in (("random_prompt", ids),
                             ("repetitive_prompt", ids[:, :64].repeat(1, max(1, a.in_len // 64))[:, :a.in_len])):
            h = Holder()
            for _ in range(a.trials + 1):
                st = NativeStepper(target, capacity=a.in_len + a.out_len + 32)
                lookup_generate(h, prompt, max_new_tokens=a.out_len, num_output_tokens=3, max_matching_ngram_size=2,
                                stepper=st)
            emit("lookup_gpu", h.first_cost, h.rest_cost_mean, prompt=name, n_drafted=h.n_drafted, n_matched=h.n_matched,
                 speedup_vs_fp16=None if base_rest is None else round(base_rest / h.rest_cost_mean, 3))


if __name__ == "__main__":
    main()
