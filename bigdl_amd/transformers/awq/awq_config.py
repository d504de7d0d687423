"""AWQ quantization config (reference: python/llm/src/ipex_llm/transformers/awq/awq_config.py)."""
from enum import Enum


class AwqBackendPackingMethod(str, Enum):
    AUTOAWQ = "autoawq"
    LLMAWQ = "llm-awq"


class AWQLinearVersion(str, Enum):
    GEMM = "gemm"
    GEMV = "gemv"

    @staticmethod
    def from_str(version: str):
        version = version.lower()
        if version == "gemm":
            return AWQLinearVersion.GEMM
        if version == "gemv":
            return AWQLinearVersion.GEMV
        raise ValueError(f"Unknown AWQLinearVersion {version}")


class AwqConfig:
    """Mirror of HF's AwqConfig fields that matter for loading: bits, group_size, zero_point, version, backend."""

    def __init__(self, bits: int = 4, group_size: int = 128, zero_point: bool = True,
                 version=AWQLinearVersion.GEMM, backend=AwqBackendPackingMethod.AUTOAWQ,
                 modules_to_not_convert=None, **kwargs):
        self.quant_method = "awq"
        self.bits = bits
        self.group_size = group_size
        self.zero_point = zero_point
        self.version = AWQLinearVersion.from_str(version) if isinstance(version, str) else version
        self.backend = AwqBackendPackingMethod(backend) if isinstance(backend, str) else backend
        self.modules_to_not_convert = modules_to_not_convert
        self.post_init()

    def post_init(self):
        if self.bits != 4:
            raise ValueError("Only 4-bit AWQ checkpoints are supported")
        if self.version not in (AWQLinearVersion.GEMM, AWQLinearVersion.GEMV):
            raise ValueError("Only gemm / gemv AWQ versions are supported")

    @classmethod
    def from_dict(cls, d, **kwargs):
        d = dict(d)
        d.pop("quant_method", None)
        return cls(**d)

    def to_dict(self):
        return {"quant_method": "awq", "bits": self.bits, "group_size": self.group_size,
                "zero_point": self.zero_point, "version": self.version.value, "backend": self.backend.value}
