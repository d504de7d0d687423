"""Pipeline-parallel stages (implementation: bigdl_amd/transformers/pipeline_parallel.py, which keeps the
reference's module path and names)."""
from ..transformers.pipeline_parallel import (PPModelWorker, init_pipeline_parallel, layer_range,  # noqa: F401
                                              pipeline_parallel, pipeline_parallel_generate)
