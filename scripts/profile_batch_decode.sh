#!/bin/bash
# Per-kernel time of batched decode steps (benchmark/batch_decode_probe.py) on the GPU box.
# Usage (from the repo root, through gpurun): bash scripts/profile_batch_decode.sh <tag> <batch> [<batch> ...]
set -e
tag=$1; shift
root=$(pwd)
export PYTHONPATH=$root
cd /tmp && export TMPDIR=/tmp
for b in "$@"; do
  rm -rf /tmp/prof_b$b
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b$b -o s -- python3 $root/benchmark/batch_decode_probe.py --batch $b --ctx 1024 --steps 32 > $root/gpurun_out/${tag}_batch$b.log 2>&1
  cp $(find /tmp/prof_b$b -name "*kernel_stats.csv" | head -1) $root/gpurun_out/${tag}_batch${b}_kernel_stats.csv
  grep ms_per_step $root/gpurun_out/${tag}_batch$b.log
done
