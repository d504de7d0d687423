#!/bin/bash
# Kernel-time and hardware-counter profiles of the flagship benchmark on the GPU box; results under gpurun_out/.
# Usage (from the repo root, through gpurun): bash scripts/profile_gpu.sh <tag>
set -e
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export PYTHONPATH=.
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > $out/bench.json 2> $out/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python bench.py --steps 1 --warmup 1 > $out/stats.log 2>&1
# counters: one pass per block budget (8 SQ; 4 TCC)
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $out/pmc_sq -o p -- python bench.py --steps 1 --warmup 0 --out-len 16 > $out/pmc_sq.log 2>&1
BIGDL_PMC_CALIBRATE=1 timeout -s KILL 400 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc -o p -- python bench.py --steps 1 --warmup 0 --out-len 16 > $out/pmc_tcc.log 2>&1
python benchmark/pmc_summary.py $out/pmc_sq $out/pmc_tcc > $out/pmc_summary.json
find $out -name "*counter_collection.csv" -size +2M -delete
find $out -name "*kernel_trace.csv" -size +8M -delete
