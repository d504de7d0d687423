"""--load-in-low-bit next to vLLM's own server arguments (reference
python/llm/src/ipex_llm/vllm/cpu/entrypoints/openai/cli_args.py)."""


def make_arg_parser(parser):
    from vllm.entrypoints.openai.cli_args import make_arg_parser as _make
    parser = _make(parser)
    parser.add_argument("--load-in-low-bit", type=str, default="sym_int4",
                        help="low-bit format for the model weights (sym_int4, asym_int4, fp8, fp6, nf4, fp16, ...)")
    return parser
