"""OpenAI-compatible vLLM server with low-bit weights:

    python -m ipex_llm.vllm.cpu.entrypoints.openai.api_server --model <path> --load-in-low-bit fp8 \\
        --tensor-parallel-size 8 --served-model-name <name> --port 8000

Reference: python/llm/src/ipex_llm/vllm/cpu/entrypoints/openai/api_server.py (a fork of vLLM's server that builds
IPEXLLMAsyncLLMEngine).  Here vLLM's own server is run unchanged after the model runners' load_model has been
rebound, which has the same effect without carrying a copy of the server."""
import sys


def main(argv=None):
    try:
        from vllm.entrypoints.openai import api_server as _server
        from vllm.utils import FlexibleArgumentParser
    except ImportError as e:                      # pragma: no cover - depends on the environment
        raise SystemExit(f"vLLM (ROCm build) is required for this entry point: {e}")
    import uvloop
    from ....model_convert import _ipex_llm_convert
    from .cli_args import make_arg_parser
    parser = make_arg_parser(FlexibleArgumentParser(description="vLLM OpenAI-compatible server with low-bit weights"))
    args = parser.parse_args(argv)
    _ipex_llm_convert(args.load_in_low_bit)
    uvloop.run(_server.run_server(args))


if __name__ == "__main__":
    main(sys.argv[1:])
