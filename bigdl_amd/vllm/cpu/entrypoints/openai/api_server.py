"""``python -m ipex_llm.vllm.cpu.entrypoints.openai.api_server ...``: the reference keeps a CPU twin of its server
(python/llm/src/ipex_llm/vllm/cpu/entrypoints/openai/api_server.py); here both import paths start the same server, see
``vllm/xpu/entrypoints/openai/api_server.py``."""
import sys

from ....xpu.entrypoints.openai.api_server import main, native_args  # noqa: F401

if __name__ == "__main__":
    main(sys.argv[1:])
