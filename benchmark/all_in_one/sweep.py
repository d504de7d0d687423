"""Runs the all-in-one protocol over a grid of (model, low_bit, batch) on one GPU and concatenates the CSVs.

    python benchmark/all_in_one/sweep.py -o gpurun_out/allinone

The grid follows the reference's CI perf configs (python/llm/test/benchmark/stable-version-arc-perf-test-sym_int4.yaml:
in/out pairs 32-32 / 1024-128 / 2048-256, batch 1/2/4/8; python/llm/dev/benchmark/all-in-one/config.yaml: 512-256)."""
import argparse
import csv
import os
import sys

import yaml

sys.path.insert(0, os.path.dirname(__file__))
import run as R  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", "--output_dir", default=".")
    ap.add_argument("--models", default="synthetic:llama2-7b,synthetic:tinyllama,synthetic:mixtral-8x7b")
    ap.add_argument("--low-bits", default="sym_int4,asym_int4,nf4,fp4,fp6,fp8_e5m2,sym_int8")
    ap.add_argument("--batches", default="1,2,4,8")
    ap.add_argument("--pairs", default="32-32,512-256,1024-128,2048-256")
    a = ap.parse_args()
    os.makedirs(a.output_dir, exist_ok=True)
    conf = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "config.yaml")))
    rows = []
    out = os.path.join(a.output_dir, "native_engine_gpu-sweep.csv")
    for model in a.models.split(","):
        lbs = a.low_bits.split(",") if model.endswith("llama2-7b") else ["sym_int4"]
        for lb in lbs:
            bss = [int(b) for b in a.batches.split(",")] if lb == "sym_int4" else [1]
            for bs in bss:
                pairs = a.pairs.split(",") if lb == "sym_int4" else ["1024-128"]
                rows += R.run_model(model, "native_engine_gpu", pairs, "", conf["warm_up"], conf["num_trials"], lb, bs, conf)
                with open(out, "w", newline="") as f:          # rewrite after every cell: progress survives a kill
                    w = csv.writer(f)
                    w.writerow(R.COLUMNS)
                    w.writerows(rows)
                print(rows[-1], flush=True)
    print(f"wrote {out}")


if __name__ == "__main__":
    main()
