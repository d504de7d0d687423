from .awq_config import AwqConfig, AwqBackendPackingMethod, AWQLinearVersion  # noqa: F401
from .linear import WQLinear_GEMM, WQLinear_GEMV  # noqa: F401
from .awq import _replace_with_awq_layers, get_layer_type, get_blocks  # noqa: F401
