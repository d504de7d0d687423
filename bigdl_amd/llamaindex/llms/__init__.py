from .bigdlllm import IpexLLM  # noqa: F401

BigdlLLM = IpexLLM
