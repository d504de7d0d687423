"""OpenAI-compatible server with low-bit weights:

    python -m ipex_llm.vllm.xpu.entrypoints.openai.api_server --model <path> --load-in-low-bit fp8 \\
        --tensor-parallel-size 8 --served-model-name <name> --port 8000

Reference: python/llm/src/ipex_llm/vllm/xpu/entrypoints/openai/api_server.py (a fork of vLLM's server that builds
IPEXLLMAsyncLLMEngine).  With vLLM installed, vLLM's own server is run unchanged after the model runners' load_model has
been rebound, which has the same effect without carrying a copy of the server.  Without vLLM (or with
``IPEX_LLM_NOT_USE_VLLM=1``) the same command line starts this package's OpenAI-compatible server
(``serving/fastapi``) on the continuous batcher: ``--model``, ``--served-model-name``, ``--host``, ``--port``,
``--load-in-low-bit``, ``--max-num-seqs`` and ``--max-model-len`` keep their meaning, other vLLM options are ignored."""
import sys


def native_args(argv):
    """The subset of vLLM's server options that the built-in server understands; the rest is returned as ignored."""
    import argparse
    ap = argparse.ArgumentParser(description="OpenAI-compatible server with low-bit weights (built-in engine)")
    ap.add_argument("--model", required=True)
    ap.add_argument("--served-model-name", default=None)
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--load-in-low-bit", default="sym_int4")
    ap.add_argument("--max-num-seqs", type=int, default=64)
    ap.add_argument("--max-model-len", type=int, default=4096)
    ap.add_argument("--tensor-parallel-size", type=int, default=1)
    return ap.parse_known_args(argv)


def main(argv=None):
    from ....engine import _have_vllm
    if not _have_vllm():
        a, ignored = native_args(argv)
        if a.tensor_parallel_size > 1:
            raise SystemExit("the built-in server shards a model with one process per GPU: "
                             "torchrun --nproc-per-node N examples/GPU/Tensor-Parallel/serving.py ...")
        if ignored:
            print(f"[bigdl_amd] vLLM options without an equivalent here, ignored: {' '.join(ignored)}", file=sys.stderr)
        from .....serving.fastapi.api_server import serve
        return serve(a.model, a.load_in_low_bit, a.port, a.host, a.served_model_name, max_num_seqs=a.max_num_seqs,
                     max_seq_len=a.max_model_len)
    from vllm.entrypoints.openai import api_server as _server
    from vllm.utils import FlexibleArgumentParser
    import uvloop
    from ....model_convert import _ipex_llm_convert
    from .cli_args import make_arg_parser
    parser = make_arg_parser(FlexibleArgumentParser(description="vLLM OpenAI-compatible server with low-bit weights"))
    args = parser.parse_args(argv)
    _ipex_llm_convert(args.load_in_low_bit)
    uvloop.run(_server.run_server(args))


if __name__ == "__main__":
    main(sys.argv[1:])
