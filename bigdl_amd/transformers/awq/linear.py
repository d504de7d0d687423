"""Containers for AutoAWQ-packed linears (reference: python/llm/src/ipex_llm/transformers/awq/linear.py:71-196).
They only hold the packed tensors while a checkpoint is being loaded; conversion re-blocks them to
asym_int4 (``convert.convert_gptq(awq=True)``), so they have no forward of their own."""
import torch
from torch import nn


class WQLinear_GEMM(nn.Module):
    def __init__(self, bits, group_size, in_features, out_features, bias, dev=None, backend="autoawq"):
        super().__init__()
        if bits != 4:
            raise NotImplementedError("Only 4-bit are supported for now.")
        self.in_features = in_features
        self.out_features = out_features
        self.bits = self.w_bit = bits
        self.group_size = group_size if group_size != -1 else in_features
        self.backend = backend
        assert in_features % self.group_size == 0 and out_features % (32 // bits) == 0
        per = 32 // bits
        if backend == "llm-awq":
            self.register_buffer("qweight", torch.zeros((out_features, in_features // per), dtype=torch.int32, device=dev))
            self.register_buffer("qzeros", torch.zeros((out_features, in_features // self.group_size // per + 1), dtype=torch.int32, device=dev))
            self.register_buffer("scales", torch.zeros((out_features, in_features // self.group_size), dtype=torch.float16, device=dev))
        else:
            self.register_buffer("qweight", torch.zeros((in_features, out_features // per), dtype=torch.int32, device=dev))
            self.register_buffer("qzeros", torch.zeros((in_features // self.group_size, out_features // per), dtype=torch.int32, device=dev))
            self.register_buffer("scales", torch.zeros((in_features // self.group_size, out_features), dtype=torch.float16, device=dev))
        if bias:
            self.register_buffer("bias", torch.zeros(out_features, dtype=torch.float16, device=dev))
        else:
            self.bias = None

    def forward(self, x):
        raise RuntimeError("WQLinear_GEMM is a load-time container; convert the model with ggml_convert_low_bit first")

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, bits={self.bits}, group_size={self.group_size}"


class WQLinear_GEMV(WQLinear_GEMM):
    pass
