#!/bin/bash
# Per-kernel time of QLoRA training steps (benchmark/qlora_bench.py) on the GPU box.
# Usage (from the repo root, through gpurun): bash scripts/profile_qlora.sh <tag> [extra qlora_bench args]
set -e
tag=$1; shift
root=$(pwd)
export PYTHONPATH=$root
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_q
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q -o s -- python3 $root/benchmark/qlora_bench.py --steps 4 "$@" > $root/gpurun_out/${tag}.log 2>&1
cp $(find /tmp/prof_q -name "*kernel_stats.csv" | head -1) $root/gpurun_out/${tag}_kernel_stats.csv
grep -h "s_per_it\|s/it" $root/gpurun_out/${tag}.log | tail -2
