"""Environment report (reference: python/llm/scripts/env-check.sh and ``ipex-llm-init``): versions, GPU, native
libraries, distributed settings."""
from __future__ import annotations

import os
import platform
import shutil
import subprocess


def collect() -> dict:
    import torch
    info = {"python": platform.python_version(), "platform": platform.platform(), "torch": torch.__version__,
            "hip": getattr(torch.version, "hip", None), "gpu_count": torch.cuda.device_count()}
    try:
        import transformers
        info["transformers"] = transformers.__version__
    except Exception as e:
        info["transformers"] = f"missing ({e})"
    if torch.cuda.device_count():
        p = torch.cuda.get_device_properties(0)
        info["gpu0"] = {"name": p.name, "arch": getattr(p, "gcnArchName", ""), "memory_gb": round(p.total_memory / 2 ** 30, 1),
                        "compute_units": p.multi_processor_count}
    from ..ops import hip, host
    info["host_library"] = str(host._LIB_PATH) if host._LIB_PATH.exists() else "not built"
    info["hip_library"] = str(hip._LIB_PATH) if hip._LIB_PATH.exists() else "not built"
    info["hipcc"] = shutil.which("hipcc") or "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None
    for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "MASTER_ADDR", "WORLD_SIZE", "IPEX_LLM_QUANTIZE_KV_CACHE",
              "IPEX_LLM_COMPRESS_KV_CACHE", "IPEX_LLM_PERFORMANCE_MODE", "IPEX_LLM_LOW_MEM", "IPEX_LLM_LAST_LM_HEAD",
              "IPEX_LLM_NOT_USE_VLLM", "IPEX_LLM_FASTAPI_TIMEOUT", "RANK_WAIT_TIME", "KV_CACHE_ALLOC_BLOCK_LENGTH",
              "BIGDL_LLM_LINEAR_THRESHOLD", "BIGDL_NATIVE_ENGINE", "BIGDL_KV_CACHE_INT4", "BIGDL_PREFILL_DENSE_CACHE",
              "BIGDL_SERVING_PIGGYBACK", "BIGDL_XGMI_ALLREDUCE", "BIGDL_HIP_ATTENTION", "BIGDL_MOE_GROUPED_GEMM"):
        if k in os.environ:
            info.setdefault("env", {})[k] = os.environ[k]
    if shutil.which("rocm-smi"):
        try:
            info["rocm_smi"] = subprocess.run(["rocm-smi", "--showproductname"], capture_output=True, text=True,
                                              timeout=20).stdout.strip().splitlines()[:8]
        except Exception:
            pass
    return info


def main():
    import json
    print(json.dumps(collect(), indent=2))


if __name__ == "__main__":
    main()
