#!/bin/bash
# Entry point of the serving image.  MODE selects the server:
#   fastapi   single GPU, OpenAI / TGI style routes with continuous batching            (default)
#   tp        tensor parallel over $GPUS GPUs of this node (one process per GPU, RCCL over xGMI)
#   pp        pipeline-parallel serving over $GPUS GPUs (layers split, micro-batches in flight)
#   fastchat  FastChat worker that registers with $CONTROLLER (+ the TGI front end on $PORT when TGI=1)
set -euo pipefail
MODEL=${MODEL:-/models/model}
LOW_BIT=${LOW_BIT:-sym_int4}
PORT=${PORT:-8000}
GPUS=${GPUS:-1}
MODE=${MODE:-fastapi}
export HSA_ENABLE_IPC_MODE_LEGACY=0
case "$MODE" in
  fastapi)
    exec python -m bigdl_amd.serving.fastapi.api_server --repo-id-or-model-path "$MODEL" --low-bit "$LOW_BIT" --port "$PORT" \
      --host 0.0.0.0 ;;
  tp)
    exec python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node "$GPUS" \
      /examples/GPU/Tensor-Parallel/serving.py --repo-id-or-model-path "$MODEL" --low-bit "$LOW_BIT" --port "$PORT" ;;
  pp)
    exec python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node "$GPUS" \
      /examples/GPU/Pipeline-Parallel-Serving/pipeline_serving.py --repo-id-or-model-path "$MODEL" --low-bit "$LOW_BIT" \
      --port "$PORT" --max-num-seqs "${MAX_NUM_SEQS:-8}" ;;
  fastchat)
    if [ "${TGI:-0}" = "1" ]; then
      python -m bigdl_amd.serving.fastchat.tgi_api_server --controller-address "${CONTROLLER:-http://localhost:21001}" \
        --host 0.0.0.0 --port "$PORT" &
    fi
    exec python -m bigdl_amd.serving.fastchat.ipex_llm_worker --model-path "$MODEL" --low-bit "$LOW_BIT" --device cuda \
      --controller-address "${CONTROLLER:-http://localhost:21001}" --worker-address "http://$(hostname -i):${WORKER_PORT:-21002}" \
      --host 0.0.0.0 --port "${WORKER_PORT:-21002}" ;;
  *) echo "unknown MODE=$MODE" >&2; exit 2 ;;
esac
