"""``python -m ipex_llm.serving.fastchat.model_worker --model-path <path containing "bigdl"> --device xpu ...``

The reference ships a copy of FastChat's ``model_worker`` that loads its model through the patched adapters
(python/llm/src/ipex_llm/serving/fastchat/model_worker.py:195-376: ``ModelWorker`` with ``generate_stream_gate``,
``generate_gate``, ``get_embeddings``).  Here FastChat's own module is run after ``patch_fastchat()`` -- same command
line, same worker, no forked copy -- and ``ModelWorker`` names the worker class of this package for programs that
construct one themselves."""
import sys

from .ipex_llm_worker import BigDLLLMWorker as ModelWorker  # noqa: F401


def main(argv=None):
    try:
        import fastchat.serve.model_worker as stock
    except ImportError as e:
        raise SystemExit("fschat is not installed: `python -m ipex_llm.serving.fastapi.api_server` is the built-in "
                         f"OpenAI / TGI compatible server ({e})")
    from .bigdl_llm_model import patch_fastchat
    patch_fastchat()
    argv = list(sys.argv[1:] if argv is None else argv)
    # the reference's worker is started with --device xpu; FastChat's argument parser knows cuda
    argv = ["cuda" if (a == "xpu" and i and argv[i - 1] == "--device") else a for i, a in enumerate(argv)]
    sys.argv = [sys.argv[0]] + argv
    import runpy
    runpy.run_module(stock.__name__, run_name="__main__")


if __name__ == "__main__":
    main()
