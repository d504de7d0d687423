"""KV caches for the module-level (HuggingFace forward) path.

Capability map to the reference python/llm/src/ipex_llm/transformers/kv.py: ``DynamicFp8Cache`` :32-75,
``DynamicNormalCache`` :78-151, ``compress_kv`` (SnapKV) :171-243, ``DynamicCompressCache`` :246-373,
``DynamicCompressFp8Cache`` :376-449.  They plug into HF's layer-wise ``Cache`` container (one layer object per
decoder layer).  The decode hot path does not use them: the native engine owns a fixed-capacity cache
(bigdl_amd/models/decoder.py).

* Normal: pre-allocated buffer that grows by ``KV_CACHE_ALLOC_BLOCK_LENGTH`` tokens (no per-token ``torch.cat``).
* Fp8: e5m2 bytes (half the memory); keys/values are returned dequantized for the attention call.
* Compress: at prefill the prompt KV is reduced to ``max_capacity_prompt`` entries chosen by the attention the last
  ``window_size`` queries pay to them (max-pooled scores + top-k), as in SnapKV; positions keep counting the
  real length.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn.functional as F
from transformers.cache_utils import Cache, DynamicLayer

KV_CACHE_ALLOC_BLOCK_LENGTH = int(os.environ.get("KV_CACHE_ALLOC_BLOCK_LENGTH", 256))


class NormalLayer(DynamicLayer):
    """[B, H, cap, D] buffers; ``update`` writes into the free tail and returns views of the filled part."""

    def lazy_initialization(self, key_states, value_states):
        self.dtype, self.device = key_states.dtype, key_states.device
        self.keys = torch.empty(0, dtype=self.dtype, device=self.device)
        self.values = torch.empty(0, dtype=self.dtype, device=self.device)
        self._kbuf = self._vbuf = None
        self._len = 0
        self.is_initialized = True

    def _reserve(self, key_states, value_states, need):
        b, h, _, d = key_states.shape
        cap = 0 if self._kbuf is None else self._kbuf.shape[2]
        if need <= cap:
            return
        new_cap = need + KV_CACHE_ALLOC_BLOCK_LENGTH
        kb = torch.empty(b, h, new_cap, d, dtype=self._store_dtype(key_states), device=self.device)
        vb = torch.empty(b, value_states.shape[1], new_cap, value_states.shape[3], dtype=self._store_dtype(value_states),
                         device=self.device)
        if self._len:
            kb[:, :, : self._len] = self._kbuf[:, :, : self._len]
            vb[:, :, : self._len] = self._vbuf[:, :, : self._len]
        self._kbuf, self._vbuf = kb, vb

    def _store_dtype(self, t):
        return t.dtype

    def _write(self, buf, start, x):
        buf[:, :, start:start + x.shape[2]] = x

    def _read(self, buf, length, dtype):
        return buf[:, :, :length]

    def update(self, key_states, value_states, *args, **kwargs):
        if not self.is_initialized:
            self.lazy_initialization(key_states, value_states)
        n = key_states.shape[2]
        self._reserve(key_states, value_states, self._len + n)
        self._write(self._kbuf, self._len, key_states)
        self._write(self._vbuf, self._len, value_states)
        self._len += n
        self.keys = self._read(self._kbuf, self._len, key_states.dtype)
        self.values = self._read(self._vbuf, self._len, value_states.dtype)
        return self.keys, self.values

    def get_seq_length(self) -> int:
        return self._len if self.is_initialized else 0

    def crop(self, tokens_to_remove=None, max_length=None):
        """Keep the first `max_length` tokens (or drop `tokens_to_remove` from the end)."""
        if max_length is None:
            max_length = self._len - (tokens_to_remove or 0)
        if max_length < 0:
            max_length = self._len + max_length
        self._len = max(0, min(self._len, max_length))
        if self._kbuf is not None:
            self.keys = self._read(self._kbuf, self._len, self.dtype)
            self.values = self._read(self._vbuf, self._len, self.dtype)


class Fp8Layer(NormalLayer):
    """e5m2 storage; the attention call gets fp16/bf16 tensors restored by the fp8 kernel."""

    def _store_dtype(self, t):
        return torch.uint8

    def _write(self, buf, start, x):
        dst = buf[:, :, start:start + x.shape[2]]
        if x.device.type == "cuda":
            from ..ops import hip
            hip.to_e5m2(x if x.stride(-1) == 1 else x.contiguous(), dst)
        else:
            dst.copy_(x.to(torch.float8_e5m2).view(torch.uint8))

    def _read(self, buf, length, dtype):
        src = buf[:, :, :length]
        if src.device.type == "cuda":
            from ..ops import hip
            out = torch.empty(src.shape, dtype=dtype if dtype in (torch.float16, torch.bfloat16, torch.float32)
                              else torch.float16, device=src.device)
            hip.from_e5m2(src, out)
            return out
        return src.contiguous().view(torch.float8_e5m2).to(dtype)


class Fp8RawLayer(Fp8Layer):
    """e5m2 storage that is never expanded: K [B, H, cap, D] and V^T [B, H, D, cap] bytes (the layout the attention
    kernels read: one launch attends to the cache bytes where they lie), and ``update`` hands back byte VIEWS
    (``uint8``: K [B, H, len, D], V^T [B, H, D, len]).  Only for models whose attention goes through the ``bigdl_hip``
    registry entry (models/hf_patch.py::hip_attention_interface), which recognises those views; the plain ``Fp8Layer``
    restores an fp16 copy of the whole cache on every step, which is what the reference's ``sdp_fp8`` avoids
    (python/llm/src/ipex_llm/transformers/models/common.py:243-253)."""

    def _reserve(self, key_states, value_states, need):
        b, h, _, d = key_states.shape
        cap = 0 if self._kbuf is None else self._kbuf.shape[2]
        if need <= cap:
            return
        new_cap = (need + KV_CACHE_ALLOC_BLOCK_LENGTH + 255) // 256 * 256
        kb = torch.zeros(b, h, new_cap, d, dtype=torch.uint8, device=self.device)
        vb = torch.zeros(b, value_states.shape[1], value_states.shape[3], new_cap, dtype=torch.uint8, device=self.device)
        if self._len:
            kb[:, :, : self._len] = self._kbuf[:, :, : self._len]
            vb[:, :, :, : self._len] = self._vbuf[:, :, :, : self._len]
        self._kbuf, self._vbuf = kb, vb

    def _views(self):
        return self._kbuf[:, :, : self._len], self._vbuf[:, :, :, : self._len]

    def update(self, key_states, value_states, *args, **kwargs):
        from ..ops import hip
        if not self.is_initialized:
            self.lazy_initialization(key_states, value_states)
        n = key_states.shape[2]
        self._reserve(key_states, value_states, self._len + n)
        hip.to_e5m2(key_states if key_states.stride(-1) == 1 else key_states.contiguous(),
                    self._kbuf[:, :, self._len:self._len + n])
        hip.to_e5m2(value_states.transpose(2, 3).contiguous(), self._vbuf[:, :, :, self._len:self._len + n])
        self._len += n
        self.keys, self.values = self._views()
        return self.keys, self.values

    def crop(self, tokens_to_remove=None, max_length=None):
        if max_length is None:
            max_length = self._len - (tokens_to_remove or 0)
        if max_length < 0:
            max_length = self._len + max_length
        self._len = max(0, min(self._len, max_length))
        if self._kbuf is not None:
            self.keys, self.values = self._views()

    def reorder_cache(self, beam_idx):
        if self._kbuf is not None:
            self._kbuf = self._kbuf.index_select(0, beam_idx.to(self._kbuf.device))
            self._vbuf = self._vbuf.index_select(0, beam_idx.to(self._vbuf.device))
            self.keys, self.values = self._views()

    def batch_repeat_interleave(self, repeats):
        if self._kbuf is not None:
            self._kbuf = self._kbuf.repeat_interleave(repeats, dim=0)
            self._vbuf = self._vbuf.repeat_interleave(repeats, dim=0)
            self.keys, self.values = self._views()

    def batch_select_indices(self, indices):
        if self._kbuf is not None:
            self._kbuf = self._kbuf[indices, ...]
            self._vbuf = self._vbuf[indices, ...]
            self.keys, self.values = self._views()


def fp8_raw_to_dense(key_bytes, vt_bytes, dtype=torch.float16):
    """(K [B,H,L,D], V^T [B,H,D,L]) byte views of an ``Fp8RawLayer`` -> dense (K, V) [B,H,L,D]: the way out for an
    attention call the byte-reading kernel does not cover."""
    from ..ops import hip
    k = torch.empty(key_bytes.shape, dtype=dtype, device=key_bytes.device)
    vt = torch.empty(vt_bytes.shape, dtype=dtype, device=vt_bytes.device)
    hip.from_e5m2(key_bytes, k)
    hip.from_e5m2(vt_bytes, vt)
    return k, vt.transpose(2, 3)


def snapkv_indices(query_states, key_states, window_size=32, max_capacity_prompt=1024, kernel_size=7, pooling="maxpool",
                   num_key_value_groups=1, scale=None):
    """SnapKV vote: which ``max_capacity_prompt - window_size`` prefix positions each kv head keeps, ascending.
    q [B,Hq,S,D], k [B,Hkv,S,D] (both already rotated) -> int64 [B,Hkv,keep]."""
    b, hq, s, d = query_states.shape
    hkv = key_states.shape[1]
    k_rep = key_states.repeat_interleave(num_key_value_groups, 1) if num_key_value_groups > 1 else key_states
    q_win = query_states[:, :, -window_size:, :].float()
    attn = torch.matmul(q_win, k_rep.float().transpose(2, 3)) * (scale if scale is not None else 1.0 / math.sqrt(d))  # [B,Hq,W,S]
    mask = torch.full((window_size, window_size), torch.finfo(attn.dtype).min, device=attn.device)
    mask = torch.triu(mask, diagonal=1)
    attn[:, :, :, -window_size:] += mask[None, None]
    attn = torch.softmax(attn, dim=-1)
    votes = attn[:, :, :, :-window_size].sum(dim=-2)                                   # [B,Hq,S-W]
    if num_key_value_groups > 1:
        votes = votes.view(b, hkv, num_key_value_groups, -1).sum(2)
    if pooling == "avgpool":
        pooled = F.avg_pool1d(votes, kernel_size=kernel_size, padding=kernel_size // 2, stride=1)
    else:
        pooled = F.max_pool1d(votes, kernel_size=kernel_size, padding=kernel_size // 2, stride=1)
    return pooled.topk(max_capacity_prompt - window_size, dim=-1).indices.sort(dim=-1).values


def compress_kv(query_states, key_states, value_states, window_size=32, max_capacity_prompt=1024, kernel_size=7,
                pooling="maxpool", num_key_value_groups=1):
    """SnapKV selection.  q [B,Hq,S,D], k/v [B,Hkv,S,D] -> k/v [B,Hkv,min(S, max_capacity_prompt),D]."""
    b, hq, s, d = query_states.shape
    if s <= max_capacity_prompt:
        return key_states, value_states
    keep = snapkv_indices(query_states, key_states, window_size, max_capacity_prompt, kernel_size, pooling,
                          num_key_value_groups)
    idx = keep.unsqueeze(-1).expand(-1, -1, -1, d)
    k_past = key_states[:, :, :-window_size, :].gather(2, idx)
    v_past = value_states[:, :, :-window_size, :].gather(2, idx)
    return (torch.cat([k_past, key_states[:, :, -window_size:, :]], dim=2),
            torch.cat([v_past, value_states[:, :, -window_size:, :]], dim=2))


class CompressLayer(NormalLayer):
    window_size = 32
    max_capacity_prompt = 1024
    kernel_size = 7
    pooling = "maxpool"

    def lazy_initialization(self, key_states, value_states):
        super().lazy_initialization(key_states, value_states)
        self.real_kv_len = 0

    def update(self, key_states, value_states, *args, query_states=None, num_key_value_groups=1, **kwargs):
        if not self.is_initialized:
            self.lazy_initialization(key_states, value_states)
        n = key_states.shape[2]
        first = self._len == 0
        self.real_kv_len += n
        if first and query_states is not None and n > self.max_capacity_prompt:
            ck, cv = compress_kv(query_states, key_states, value_states, self.window_size, self.max_capacity_prompt,
                                 self.kernel_size, self.pooling, num_key_value_groups)
            super().update(ck, cv)
            return key_states, value_states      # the prefill itself attends to the full prompt
        return super().update(key_states, value_states)

    def get_seq_length(self) -> int:
        return self.real_kv_len if self.is_initialized else 0

    def get_mask_sizes(self, query_length: int):
        stored = self._len if self.is_initialized else 0
        if stored == 0:
            return query_length, 0
        return stored + query_length, 0


class CompressFp8Layer(CompressLayer, Fp8Layer):
    pass


class DynamicNormalCache(Cache):
    def __init__(self, *args, **kwargs):
        super().__init__(layer_class_to_replicate=NormalLayer)


class DynamicFp8Cache(Cache):
    def __init__(self, *args, **kwargs):
        super().__init__(layer_class_to_replicate=Fp8Layer)


class DynamicFp8RawCache(DynamicFp8Cache):
    """fp8 cache whose layers hand the attention function byte views (see ``Fp8RawLayer``)"""

    def __init__(self, *args, **kwargs):
        Cache.__init__(self, layer_class_to_replicate=Fp8RawLayer)


class DynamicCompressCache(Cache):
    wants_query = True

    def __init__(self, *args, window_size=32, max_capacity_prompt=1024, kernel_size=7, pooling="maxpool", **kwargs):
        cls = type("CompressLayerCfg", (self._layer_base(),), dict(window_size=window_size, kernel_size=kernel_size,
                                                                     max_capacity_prompt=max_capacity_prompt,
                                                                     pooling=pooling))
        super().__init__(layer_class_to_replicate=cls)

    @staticmethod
    def _layer_base():
        return CompressLayer

    @property
    def real_kv_len(self):
        return self.layers[0].real_kv_len if self.layers else 0


class DynamicCompressFp8Cache(DynamicCompressCache):
    @staticmethod
    def _layer_base():
        return CompressFp8Layer


def make_kv_cache(quantize_kv=False, compress_kv_=False, raw_fp8=False, **kwargs) -> Cache:
    """Cache selection used by the bound model forwards (reference models/llama.py:73-94).  ``raw_fp8``: the model's
    attention reads e5m2 bytes itself (``DynamicFp8RawCache``)."""
    if quantize_kv and raw_fp8 and not compress_kv_:
        return DynamicFp8RawCache()
    if compress_kv_:
        return DynamicCompressFp8Cache(**kwargs) if quantize_kv else DynamicCompressCache(**kwargs)
    return DynamicFp8Cache() if quantize_kv else DynamicNormalCache()
