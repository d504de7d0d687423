from .vllm import VLLM  # noqa: F401

__all__ = ["VLLM"]
