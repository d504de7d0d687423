"""Multi-GPU execution: one process per GPU, torch.distributed over RCCL / xGMI.

* ``comm``             process-group bootstrap, (dp, pp, tp) grid, bucketed all-reduce
* ``tensor_parallel``  column / row sharding of quantized weights (HF module path and native engine)
* ``pipeline``         layer-range pipeline stages + pipelined generation (re-exported by
                       ``bigdl_amd.transformers.pipeline_parallel`` under the reference's names)
* ``data_parallel``    gradient synchronisation for LoRA / QLoRA fine-tuning
"""
from .comm import all_reduce_, all_reduce_coalesced_, barrier, init_distributed, new_groups, rank, world_size  # noqa: F401

from .expert_parallel import shard_engine_experts  # noqa: E402,F401
