#!/bin/bash
# One-shot benchmark inside the inference image: flagship decode benchmark + the all-in-one table for config.yaml.
set -euo pipefail
cd /opt/bigdl_amd
python bench.py --steps "${STEPS:-5}" --warmup "${WARMUP:-2}" --gpus "${GPUS:-1}" ${BENCH_ARGS:-}
cd /benchmark/all_in_one && python run.py -o "${OUT:-/tmp/results}"
