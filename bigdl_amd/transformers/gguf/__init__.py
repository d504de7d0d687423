from .api import load_gguf_model  # noqa: F401
from .gguf import GGUFFileLoader, GGUFWriter  # noqa: F401
