"""TGI-compatible front end (reference: python/llm/src/ipex_llm/serving/fastchat/tgi_api_server.py).  The route set
(``/generate``, ``/generate_stream``) is served by the built-in app in ``bigdl_amd.serving.fastapi``."""
from ..fastapi.api_server import main, serve  # noqa: F401

if __name__ == "__main__":
    main()
