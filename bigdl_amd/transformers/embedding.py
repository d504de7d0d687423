"""Embedding variants (reference: python/llm/src/ipex_llm/transformers/embedding.py:29-241).

* ``CPUEmbedding``: table pinned in host memory, lookups on the host, result moved to the device.  On a
  288 GB MI355X this is never needed for capacity; it is kept because the kwarg is part of the API.
* ``DiskEmbedding``: fp16 table in a file, one seek per token.
* ``LowBitEmbedding``: block-quantized rows (any row format of the host library, e.g. q2_k / q4_k /
  sym_int4), gathered + dequantized by ``dequantize_rows`` (HIP kernel on the device, C++ on the host).
"""
from __future__ import annotations

import os
import tempfile

import numpy as np
import torch
from torch import Tensor
from torch.nn import functional as F

from .. import qtypes as Q
from ..ops import host
from ..utils.common import invalidInputError


class CPUPinnedParam(torch.nn.Parameter):
    """Parameter that refuses to leave the host (``.to(device)`` only remembers the target)."""

    def to(self, *args, **kwargs):
        device, dtype, non_blocking, _ = torch._C._nn._parse_to(*args, **kwargs)
        if device is None or device.type == "cpu":
            return super().to(*args, **kwargs)
        return super().to(dtype=dtype) if dtype is not None else self

    def cuda(self, *a, **k):
        return self


class CPUEmbedding(torch.nn.Embedding):
    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, max_norm=None, norm_type=2.0,
                 scale_grad_by_freq=False, sparse=True, _weight=None, _freeze=False, device=None, dtype=None):
        super().__init__(num_embeddings, embedding_dim, padding_idx, max_norm, norm_type, scale_grad_by_freq, sparse,
                         _weight, True, device, dtype)
        self.weight = CPUPinnedParam(self.weight.data, requires_grad=not _freeze)
        self.out_device = None

    def _apply(self, fn, recurse=True):
        # remember where the rest of the model went, keep the table on the host
        try:
            probe = fn(torch.empty(0))
            if probe.device.type != "cpu":
                self.out_device = probe.device
        except Exception:
            pass
        return self

    def forward(self, x: Tensor):
        dev = self.out_device or x.device
        return super().forward(x.to("cpu")).to(dev)

    @classmethod
    def from_embedding(cls, embedding: torch.nn.Embedding):
        return cls(embedding.num_embeddings, embedding.embedding_dim, embedding.padding_idx, embedding.max_norm,
                   embedding.norm_type, embedding.scale_grad_by_freq, embedding.sparse, embedding.weight.data, True,
                   embedding.weight.device, embedding.weight.dtype)


class DiskEmbedding(torch.nn.Embedding):
    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, max_norm=None, norm_type=2.0,
                 scale_grad_by_freq=False, sparse=False, _weight=None, _freeze=False, device=None, dtype=None):
        super().__init__(num_embeddings, embedding_dim, padding_idx, max_norm, norm_type, scale_grad_by_freq, sparse,
                         _weight, True, device, dtype)
        self.filename = os.path.join(tempfile.mkdtemp(prefix="bigdl_disk_embedding_"), "embeddings.bin")
        self.weight.data.flatten().to(torch.float16).numpy().tofile(self.filename)
        dummy = torch.empty(0, 0, dtype=self.weight.dtype, device=self.weight.device)
        self.weight = torch.nn.Parameter(dummy, requires_grad=False)
        self.out_device = None

    def _apply(self, fn, recurse=True):
        try:
            probe = fn(torch.empty(0))
            self.out_device = probe.device
        except Exception:
            pass
        return self

    def forward(self, input_ids: Tensor):
        ids = input_ids.cpu().flatten()
        rows = np.memmap(self.filename, dtype=np.float16, mode="r", shape=(self.num_embeddings, self.embedding_dim))
        emb = torch.from_numpy(np.ascontiguousarray(rows[ids.numpy()]))
        emb = emb.view(*input_ids.size(), self.embedding_dim)
        return emb.to(self.out_device or input_ids.device)

    @classmethod
    def from_embedding(cls, embedding: torch.nn.Embedding):
        return cls(embedding.num_embeddings, embedding.embedding_dim, embedding.padding_idx, embedding.max_norm,
                   embedding.norm_type, embedding.scale_grad_by_freq, embedding.sparse, embedding.weight.data, True,
                   embedding.weight.device, embedding.weight.dtype)

    def to_embedding(self):
        with open(self.filename, "rb") as f:
            buf = f.read()
        w = torch.frombuffer(bytearray(buf), dtype=torch.float16).clone().view(self.num_embeddings, self.embedding_dim)
        return torch.nn.Embedding(self.num_embeddings, self.embedding_dim, self.padding_idx, self.max_norm,
                                  self.norm_type, self.scale_grad_by_freq, self.sparse, w, True)

    @staticmethod
    def replace_normal_embedding(m: torch.nn.Module):
        for name, module in m.named_children():
            if type(module) == torch.nn.Embedding:
                m._modules[name] = DiskEmbedding.from_embedding(module)
            else:
                DiskEmbedding.replace_normal_embedding(module)

    @staticmethod
    def restore_normal_embedding(m: torch.nn.Module):
        for name, module in m.named_children():
            if type(module) == DiskEmbedding:
                m._modules[name] = module.to_embedding()
            else:
                DiskEmbedding.restore_normal_embedding(module)


class LowBitEmbedding(torch.nn.Embedding):
    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, max_norm=None, norm_type=2.0,
                 scale_grad_by_freq=False, sparse=False, _weight=None, _freeze=False, device=None, dtype=None,
                 convert_shape_only=None, qtype=None):
        super().__init__(num_embeddings, embedding_dim, padding_idx, max_norm, norm_type, scale_grad_by_freq, sparse,
                         _weight, True, device, dtype)
        from .low_bit_linear import FP4Params
        self.qweight = FP4Params(self.weight.data, requires_grad=False, quantized=False, _shape=None,
                                 convert_shape_only=convert_shape_only, qtype=qtype, in_features=embedding_dim,
                                 host_only=True)
        # rows are gathered individually: keep the host (row-block) layout on every device
        self.qweight.quantize("meta" if self.weight.device.type == "meta" else "cpu")
        self.qweight.requires_grad_(False)
        self.weight = torch.nn.Parameter(torch.empty(0, 0, dtype=torch.float16), requires_grad=False)
        self.embedding_dim = embedding_dim
        self.qtype = qtype

    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0))
        if probe.device != self.qweight.data.device:
            from .low_bit_linear import FP4Params
            q = self.qweight
            self.qweight = FP4Params(q.data.to(probe.device), quantized=True, _shape=q._shape, qtype=q.qtype,
                                     host_only=True)
        return self

    def forward(self, x: Tensor):
        invalidInputError(self.qweight.quantized, "LowBitEmbedding weight is not quantized")
        raw = self.qweight.data.view(torch.uint8) if self.qweight.data.dtype != torch.uint8 else self.qweight.data
        if raw.device.type == "cuda":
            from ..ops import hip
            return hip.dequant_rows(raw, self.qtype, self.num_embeddings, self.embedding_dim, ids=x.to(raw.device),
                                    dtype=torch.float32)
        return host.dequantize_rows(x, raw, self.qtype, self.embedding_dim, self.num_embeddings)

    @classmethod
    def from_embedding(cls, embedding: torch.nn.Embedding, convert_shape_only, qtype):
        return cls(embedding.num_embeddings, embedding.embedding_dim, embedding.padding_idx, embedding.max_norm,
                   embedding.norm_type, embedding.scale_grad_by_freq, embedding.sparse, embedding.weight.data, True,
                   embedding.weight.device, embedding.weight.dtype, convert_shape_only, qtype)
