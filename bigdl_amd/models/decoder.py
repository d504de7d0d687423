"""Native decoder engine for MI355X.

One configurable implementation covers the decoder-only families the reference patches one by one
(llama/mistral/qwen2/gemma/gemma2/phi3/yi/... see bigdl_amd/models/arch.py; reference files
python/llm/src/ipex_llm/transformers/models/{llama,mistral,qwen2,gemma,gemma2,phi3,...}.py).  Where the
reference rebinds HF forwards and grows its KV cache with as_strided tricks
(python/llm/src/ipex_llm/transformers/kv.py:78-151), this engine owns

* a fixed-capacity KV cache (K row-major, V transposed; fp16 or fp8-e5m2 bytes) sized up front — 288 GB of
  HBM makes capacity planning trivial, and fixed shapes are what hipGraph capture needs;
* a decode step of 5 fused launches per layer (norm+qkv GEMV, rope+append, split-KV attention,
  o_proj+residual, norm+gate/up+act GEMV, down+residual), captured once per (batch, q_len) into a hipGraph
  with the sequence length read from device memory, so a token costs one graph launch;
* an eager prefill that dequantizes weights tile-wise and runs the dense hipBLASLt GEMM + flash SDPA.
"""
from __future__ import annotations

import math
import os
import time
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn.functional as F

from .. import qtypes as Q
from ..transformers.low_bit_linear import FP4Params, GEMM_MAX_ROWS, GEMV_MAX_ROWS, dequant_weight


@dataclass
class DecoderConfig:
    vocab_size: int
    hidden_size: int
    intermediate_size: int
    num_layers: int
    num_heads: int
    num_kv_heads: int
    head_dim: int
    rms_eps: float = 1e-6
    rope_theta: float = 10000.0
    rope_scaling: Optional[dict] = None
    max_position: int = 4096
    tie_word_embeddings: bool = False
    act: str = "silu"
    norm_offset: float = 0.0          # gemma: weight is stored as (w - 1)
    embed_scale: float = 1.0          # gemma: sqrt(hidden)
    attn_scale: Optional[float] = None
    attn_softcap: float = 0.0
    final_softcap: float = 0.0
    sliding_window: int = 0
    sliding_pattern: int = 0          # gemma2: every `pattern`-th layer is global; 0 = all layers alike
    rot_dim: Optional[int] = None     # partial rotary
    rope_interleaved: bool = False
    alibi: bool = False               # baichuan-13B / bloom / mpt: per-head linear position bias instead of RoPE
    alibi_slopes: Optional[tuple] = None   # explicit per-head slopes (mpt's ordering, falcon's 1/sqrt(d) scaling)
    norm: str = "rms"                 # "layer": LayerNorm with bias (gpt-neox, bloom, mpt, falcon, gpt2, opt, phi, ...)
    gated_mlp: bool = True            # False: down(act(up(x)))
    parallel_residual: bool = False   # h + attn(ln1(h)) + mlp(ln2(h))  (gpt-neox, falcon, phi, gpt-j)
    no_rope: bool = False             # learned absolute positions (gpt2, gpt_bigcode, opt) or ALiBi
    pos_offset: int = 0               # opt: the position table starts at index 2
    post_norms: bool = False          # gemma2: extra norms after attention / mlp
    model_type: str = "llama"
    # multi-head latent attention (MiniCPM3, DeepSeek-V2): kv_lora > 0 switches it on
    mla_q_lora: int = 0               # 0: one q projection
    mla_kv_lora: int = 0
    # MoE
    num_experts: int = 0
    top_k: int = 0
    moe_intermediate_size: int = 0
    shared_expert_intermediate_size: int = 0
    norm_topk_prob: bool = True
    moe_n_group: int = 1              # DeepSeek-V2 group-limited routing: expert groups, and how many of them compete
    moe_topk_group: int = 1
    moe_score: str = "softmax"        # "sigmoid": DeepSeek-V3 (scores = sigmoid(logits), selection on scores + a learned bias)
    moe_group_score: str = "max"      # how a group is ranked: its best expert, or "top2sum" (DeepSeek-V3)
    logit_scale: float = 1.0
    # Qwen2-VL multimodal rotary embedding: how many of the rot/2 frequencies take their angle from the temporal,
    # height and width position of a token (text tokens carry the same value on all three axes)
    mrope_section: Optional[tuple] = None

    @property
    def q_size(self):
        return self.num_heads * self.head_dim

    @property
    def kv_size(self):
        return self.num_kv_heads * self.head_dim


_ACT_FNS = {"silu": F.silu, "swish": F.silu, "gelu": F.gelu, "relu": F.relu,
            "gelu_tanh": lambda x: F.gelu(x, approximate="tanh"), "gelu_pytorch_tanh": lambda x: F.gelu(x, approximate="tanh"),
            "gelu_new": lambda x: F.gelu(x, approximate="tanh"), "gelu_fast": lambda x: F.gelu(x, approximate="tanh")}


def alibi_slopes(n_heads: int) -> torch.Tensor:
    """fp32 [n_heads] ALiBi slopes: the geometric sequence 2^(-8 i / n) for a power-of-two head count, interleaved
    with the odd terms of the next power of two otherwise (Baichuan-13B ``_get_interleave``, bloom
    ``build_alibi_tensor``; reference models/baichuan.py:289-330, models/bloom.py:60-120)."""
    def pow2(n):
        start = 2.0 ** (-(2.0 ** -(math.log2(n) - 3)))
        return [start * start ** i for i in range(n)]
    if math.log2(n_heads).is_integer():
        v = pow2(n_heads)
    else:
        c = 2 ** math.floor(math.log2(n_heads))
        v = pow2(c) + pow2(2 * c)[0::2][: n_heads - c]
    return torch.tensor(v, dtype=torch.float32)


SUPPORTED_ROPE_TYPES = (None, "default", "linear", "llama3", "yarn", "longrope", "su")


def compute_inv_freq(cfg: DecoderConfig, device=None, seq_len: Optional[int] = None):
    """inv_freq (fp32 [rot/2]) and the cos/sin scaling factor for the rope variants llama3 / linear / yarn / longrope.
    ``seq_len``: the longest position the table will serve; longrope switches from its short to its long factors once
    that exceeds ``original_max_position_embeddings`` (HF modeling_rope_utils._compute_longrope_parameters)."""
    rot = cfg.rot_dim or cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
    scaling = 1.0
    rs = cfg.rope_scaling or {}
    kind = rs.get("rope_type", rs.get("type"))
    if kind == "linear":
        inv = inv / float(rs["factor"])
    elif kind == "llama3":
        factor, lo, hi = rs["factor"], rs.get("low_freq_factor", 1.0), rs.get("high_freq_factor", 4.0)
        old = rs.get("original_max_position_embeddings", 8192)
        wavelen = 2 * math.pi / inv
        smooth = ((old / wavelen) - lo) / (hi - lo)
        scaled = torch.where(wavelen > old / lo, inv / factor, inv)
        mid = (wavelen <= old / lo) & (wavelen >= old / hi)
        inv = torch.where(mid, (1 - smooth) * inv / factor + smooth * inv, scaled)
    elif kind == "yarn":
        factor = rs["factor"]
        old = rs.get("original_max_position_embeddings", cfg.max_position)
        beta_fast, beta_slow = rs.get("beta_fast", 32), rs.get("beta_slow", 1)

        def corr(n):
            return rot * math.log(old / (n * 2 * math.pi)) / (2 * math.log(cfg.rope_theta))
        low, high = max(math.floor(corr(beta_fast)), 0), min(math.ceil(corr(beta_slow)), rot - 1)
        ramp = torch.clamp((torch.arange(rot // 2, dtype=torch.float32) - low) / max(high - low, 1e-3), 0, 1)
        inv = inv / factor * ramp + inv * (1 - ramp)
        scaling = rs.get("attention_factor") or (0.1 * math.log(factor) + 1.0)
    elif kind in ("longrope", "su"):
        orig = rs.get("original_max_position_embeddings") or cfg.max_position
        factor = cfg.max_position / orig if rs.get("original_max_position_embeddings") else float(rs.get("factor", 1.0))
        use_long = seq_len is not None and seq_len > orig and "long_factor" in rs
        ext = torch.tensor(rs["long_factor"] if use_long else rs["short_factor"], dtype=torch.float32)
        inv = inv / ext
        scaling = rs.get("attention_factor") or (1.0 if factor <= 1.0 else math.sqrt(1.0 + math.log(factor) / math.log(orig)))
    elif kind not in (None, "default"):
        raise NotImplementedError(f"rope scaling type {kind!r} is not implemented on the native engine")
    return inv.to(device) if device is not None else inv, scaling


class Lin:
    """One linear layer as the engine sees it: a T16 low-bit weight (fused kernels) or a dense fp16 weight."""

    def __init__(self, weight, bias=None, out_features=None, in_features=None):
        self.bias = None if bias is None else bias.detach().to(torch.float16).contiguous()
        if isinstance(weight, FP4Params) and weight.qlayout == "t16":
            self.fused = True
            self.w = weight.data
            self.qtype = weight.qtype
            self.n, self.k = weight._shape
            self.param = weight
        else:
            self.fused = False
            if isinstance(weight, FP4Params):
                self.param = weight
                self.w = dequant_weight(weight, torch.float16)
            else:
                self.param = None
                self.w = weight.detach().to(torch.float16)
            self.qtype = Q.FP16
            self.n, self.k = self.w.shape

    @property
    def nbytes(self):
        return self.w.numel() * self.w.element_size()

    def ln_fold(self, gamma, beta):
        """(W @ gamma, W @ beta) for the LayerNorm-folding GEMV, computed once per (weight, norm) pair."""
        key = (gamma.data_ptr(), None if beta is None else beta.data_ptr())
        if getattr(self, "_ln_key", None) != key:
            from ..ops import hip
            dense, keep = Lin.cache_dense, getattr(self, "_dense", None)
            Lin.cache_dense = False                       # a transient dequantized copy, not the resident one
            try:
                self._ln, self._ln_key = hip.layer_norm_fold(self.dense(), gamma, beta), key
            finally:
                Lin.cache_dense = dense
                if keep is not None:
                    self._dense = keep
        return self._ln

    # opt-in (NativeDecoder.dense_prefill_cache / BIGDL_PREFILL_DENSE_CACHE=1): keep the fp16 form of every weight
    # resident next to the low-bit one, so that large-M prefill GEMMs skip the dequantization pass (3.7 ms of an 18.6 ms
    # first token for Llama-2-7B) at the price of 2 bytes per weight of HBM -- affordable on a 288 GB GPU for serving
    # deployments that are prefill bound; decode keeps streaming the low-bit tiles
    cache_dense = False

    def dense(self, dtype=torch.float16):
        if self.fused:
            from ..ops import hip
            if Lin.cache_dense and dtype == torch.float16:
                d = getattr(self, "_dense", None)
                if d is None:
                    d = self._dense = hip.dequant_t16(self.w, self.qtype, self.n, self.k, dtype=dtype)
                return d
            return hip.dequant_t16(self.w, self.qtype, self.n, self.k, dtype=dtype)
        return self.w

    def fp8(self):
        """(codes ``float8_e4m3fn`` [n, k], scales fp32 [n, 1]) with ``W ~= codes * scales`` (one scale per output row),
        made once and kept: the weight operand of the opt-in fp8 prompt pass (1 byte per weight of HBM on top of the
        low-bit tiles, which decode keeps using)."""
        w8 = getattr(self, "_fp8", None)
        if w8 is None:
            from ..ops import hip
            d = hip.dequant_t16(self.w, self.qtype, self.n, self.k, dtype=torch.float16) if self.fused else self.w
            sw = d.abs().amax(dim=1, keepdim=True).float().clamp_min(1e-12) / 448.0
            w8 = self._fp8 = ((d.float() / sw).to(torch.float8_e4m3fn), sw)
        return w8


@dataclass
class MLAWeights:
    """Projections of multi-head latent attention (reference models/minicpm3.py, deepseek.py): hidden -> q latent ->
    q heads, hidden -> (kv latent | shared rotary key) -> k / v heads.  The engine expands the latent into ordinary
    per-head K / V rows of ``head_dim`` channels laid out [rotary | no-position | zero padding] (values: [v | zero
    padding]), so caches and attention kernels are the standard ones; ``q_b`` / ``kv_b`` rows are stored in that order
    with zero rows at the padding and at the K rotary slots, which receive the shared rotary key."""
    a: "Lin"                 # hidden -> [q latent (q_lora, may be 0) | kv latent (kv_lora) | shared rotary key]: one GEMV
    kv_a_norm: torch.Tensor
    kv_b: "Lin"
    kv_lora: int
    q_lora: int = 0
    q: Optional["Lin"] = None            # without a q latent: hidden -> q heads
    q_a_norm: Optional[torch.Tensor] = None
    q_b: Optional["Lin"] = None

    def lins(self):
        return [l for l in (self.a, self.kv_b, self.q, self.q_b) if l is not None]


@dataclass
class LayerWeights:
    input_norm: torch.Tensor
    qkv: Lin
    o: Lin
    post_norm: Optional[torch.Tensor]      # None: the MLP reads the input norm's output (shared LayerNorm)
    gate: Optional[Lin] = None
    up: Optional[Lin] = None
    down: Optional[Lin] = None
    pre_ffn_norm: Optional[torch.Tensor] = None      # gemma2
    post_attn_out_norm: Optional[torch.Tensor] = None
    post_ffn_out_norm: Optional[torch.Tensor] = None
    moe: Optional[object] = None
    window: int = 0
    input_norm_b: Optional[torch.Tensor] = None      # LayerNorm biases
    post_norm_b: Optional[torch.Tensor] = None
    mla: Optional[MLAWeights] = None                 # latent attention: replaces qkv (which is then None)


class KVCache:
    """Fixed-capacity cache: K [L][B,Hkv,Lmax,D], V^T [L][B,Hkv,D,Lmax]; fp16, e5m2 bytes (``fp8=True``) or int4 with
    fp16 scales per key and 32 channels (``fp8="int4"``, layout in ops/hip.py::kv_int4_shapes)."""

    def __init__(self, cfg: DecoderConfig, batch: int, capacity: int, fp8, device):
        capacity = (capacity + 255) // 256 * 256
        self.mode = fp8                      # what ensure_cache compares: False / True / "int4"
        self.int4 = fp8 == "int4"
        self.fp8 = bool(fp8) and not self.int4
        self.quantized = self.fp8 or self.int4
        self.capacity = capacity
        self.batch = batch
        self.k_scale = self.v_scale = None
        if self.int4:
            from ..ops import hip
            if cfg.head_dim not in hip.KV_INT4_HEAD_DIMS:
                raise ValueError(f"the int4 KV cache needs head_dim in {hip.KV_INT4_HEAD_DIMS}, got {cfg.head_dim}")
            sk, sv, ssk, ssv = hip.kv_int4_shapes(batch, cfg.num_kv_heads, capacity, cfg.head_dim)
            self.k = [torch.zeros(sk, dtype=torch.uint8, device=device) for _ in range(cfg.num_layers)]
            self.v = [torch.zeros(sv, dtype=torch.uint8, device=device) for _ in range(cfg.num_layers)]
            self.k_scale = [torch.zeros(ssk, dtype=torch.float16, device=device) for _ in range(cfg.num_layers)]
            self.v_scale = [torch.zeros(ssv, dtype=torch.float16, device=device) for _ in range(cfg.num_layers)]
        else:
            dt = torch.uint8 if self.fp8 else torch.float16
            self.k = [torch.zeros(batch, cfg.num_kv_heads, capacity, cfg.head_dim, dtype=dt, device=device)
                      for _ in range(cfg.num_layers)]
            self.v = [torch.zeros(batch, cfg.num_kv_heads, cfg.head_dim, capacity, dtype=dt, device=device)
                      for _ in range(cfg.num_layers)]
        self.past_len = torch.zeros(batch, dtype=torch.int32, device=device)
        self.host_len = 0  # host mirror of past_len (uniform across the batch)
        # continuous batching: every batch row is a slot with its own length; rows whose `active` entry is 0 do not
        # advance when a decode step runs (their output is ignored)
        self.active = torch.ones(batch, dtype=torch.int32, device=device)
        self.active_host = [1] * batch
        self.slot_len = None  # list of per-slot lengths on the host once slot mode is on

    def reset(self):
        self.past_len.zero_()
        self.host_len = 0
        self.active.fill_(1)
        self.active_host = [1] * self.batch
        self.slot_len = None

    def enable_slots(self):
        """Switch to per-slot bookkeeping: all slots empty and inactive."""
        self.past_len.zero_()
        self.active.zero_()
        self.active_host = [0] * self.batch
        self.host_len = 0
        self.slot_len = [0] * self.batch

    def set_active(self, slot: int, on: bool):
        self.active[slot] = 1 if on else 0
        self.active_host[slot] = 1 if on else 0
        if not on:
            self.past_len[slot] = 0
            self.slot_len[slot] = 0

    def crop(self, length: int):
        """Drop everything after `length` tokens (speculative rejection): just a counter update."""
        self.past_len.fill_(length)
        self.host_len = length

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.k + self.v + (self.k_scale or []) + (self.v_scale or []))

    def scales(self, li):
        """keyword arguments of ``hip.attn_decode`` for layer ``li`` (empty unless int4)"""
        return dict(k_scale=self.k_scale[li], v_scale=self.v_scale[li]) if self.int4 else {}



def attn_decode_splits(batch: int, num_kv_heads: int, capacity: int) -> int:
    """Key ranges that one (sequence, KV head) pair of the decode attention is cut into.

    One workgroup serves one pair and one range, so a lone sequence needs ranges to fill the chip: as many as keep the
    grid within 256 workgroups, never shorter than 256 keys.  The partial results are merged by every workgroup of the
    output projection, which costs rows x ranges x q_size floats per workgroup, so from 64 pairs on a range is at least
    1024 keys long.  Measured on one MI355X at 1024 keys of context, Llama-2-7B sym_int4 + fp8 cache: 5 sequences
    2.97 -> 2.34 ms per step, 6 3.06 -> 2.45, 7 3.34 -> 2.73 with one range instead of two; at 4096 keys 2 sequences
    are fastest with 4 ranges, 3-4 with 2, 6 and more with 1.  ``BIGDL_ATTN_SPLITS`` overrides the count."""
    pairs = max(1, batch * num_kv_heads)
    max_splits = max(1, capacity // 256)
    if os.environ.get("BIGDL_ATTN_SPLITS"):
        return max(1, min(max_splits, int(os.environ["BIGDL_ATTN_SPLITS"])))
    if pairs >= 64:
        max_splits = max(1, capacity // 1024)
    return max(1, min(max_splits, 256 // pairs, 16))


class NativeDecoder:
    def __init__(self, cfg: DecoderConfig, embed: torch.Tensor, layers: list, final_norm: torch.Tensor, lm_head: Lin,
                 device="cuda", final_norm_b=None, pos_embed=None, embed_norm=None):
        self.cfg = cfg
        self.device = torch.device(device)
        self.embed = embed
        self.layers = layers
        self.cache_layer = None            # layer -> cache row index, when this engine runs a subset of another's layers
        self.final_norm = final_norm
        self.final_norm_b = final_norm_b
        self.pos_embed = pos_embed              # [max_pos (+ offset), hidden] learned absolute positions
        self.embed_norm = embed_norm            # (weight, bias) LayerNorm right after the embedding (bloom)
        self.lm_head = lm_head
        inv, scaling = compute_inv_freq(cfg, self.device)
        self.inv_freq = inv
        self.rope_scaling_factor = scaling
        self.cache: Optional[KVCache] = None
        self._graphs = {}
        self._prefill_graphs = {}          # short prompts: one captured prompt pass per length bucket (see _prefill_short)
        self._static = {}
        # prompt GEMMs above GEMM_MAX_ROWS rows run on the fused prompt-sized kernel (csrc/hip/gemm_prompt_kernel.h): no
        # fp16 scratch copy of a weight and no library GEMM.  ``BIGDL_PREFILL_LIBRARY=1`` selects dequantize + hipBLASLt
        # instead (A/B measurements).
        self.prefill_fused = os.environ.get("BIGDL_PREFILL_LIBRARY", "0") != "1"
        # ``BIGDL_DEQUANT_AHEAD=1``: while the library GEMM of one weight runs, the next weight of the prompt pass is
        # expanded to fp16 on a second stream into the other of two scratch buffers.  Measured on Llama-2-7B at 1024
        # prompt tokens: first token 19.4 against 17.5 ms (18.8 with the side stream at equal priority on another box):
        # the expansion's workgroups take CU time and cache from the GEMM they run beside, which costs more than the
        # 3.7 ms of expansion that leave the critical path.  Off by default.
        self.dequant_ahead = os.environ.get("BIGDL_DEQUANT_AHEAD", "0") == "1"
        self._ahead = None
        # opt-in fp8 prompt pass (``BIGDL_PREFILL_FP8=1``): see _linear_dense
        self.prefill_fp8 = os.environ.get("BIGDL_PREFILL_FP8") == "1"
        self._ssq_ring = {}             # rows -> [slots, rows] fp32 (see _ssq_reset)
        self._ssq_owner = None
        self._ssq_next = 0
        self.fold_batched_norm = False
        self.attn_scale = cfg.attn_scale if cfg.attn_scale is not None else 1.0 / math.sqrt(cfg.head_dim)
        # ALiBi slopes of this rank's heads (tensor parallelism re-slices them)
        self.alibi = None
        if cfg.alibi:
            sl = torch.tensor(cfg.alibi_slopes, dtype=torch.float32) if cfg.alibi_slopes else alibi_slopes(cfg.num_heads)
            self.alibi = sl.to(self.device).contiguous()
        self.use_graph = True
        # pipeline stages (bigdl_amd/parallel/pipeline.py): a stage that is not the first takes hidden states instead of
        # token ids, a stage that is not the last returns hidden states instead of logits
        self.stage_first = True
        self.stage_last = True
        # SnapKV prompt compression (transformers/kv.py ``snapkv_indices``; reference models/llama.py:182-188 +
        # transformers/kv.py:171-243): dict(window_size, max_capacity_prompt, kernel_size, pooling) or None.  After a
        # compressed prompt the cache is shorter than the sequence: decode steps take their rotary angles from a copy of
        # the table shifted by the difference (``rope_dec``), so positions keep counting from the real length.
        self.compress_kv = None
        self.rope_dec = None
        self._rope_shift = 0
        self._rope_shift_buf = None
        if os.environ.get("BIGDL_PREFILL_DENSE_CACHE") == "1":
            Lin.cache_dense = True
        # tensor parallelism (bigdl_amd/parallel/tensor_parallel.py ``shard_engine``): partial sums of the
        # row-parallel projections are all-reduced over this group
        self.tp_group = None
        self.tp_world = 1
        self.tp_fast_allreduce = None   # parallel/xgmi_allreduce.py ``XgmiAllReduce`` once its self-check passed
        # timing attributes with the reference's names (utils/benchmark_util_4_47.py:3372-3393)
        self.first_cost = self.rest_cost_mean = self.encoder_time = self.peak_memory = 0.0
        self.first_token_time = 0.0
        self.last_token_time = []

    # ------------------------------------------------------------------ cache
    def ensure_cache(self, batch: int, needed: int, fp8: bool = False):
        c = self.cache
        if c is None or c.batch != batch or c.capacity < needed or c.mode != fp8:
            cap = max(needed, 256)
            self.cache = KVCache(self.cfg, batch, cap, fp8, self.device)
            self._graphs.clear()
            self._prefill_graphs.clear()
            self._static.clear()
            # (longrope picks its factor set by the length the table has to serve)
            self.inv_freq, self.rope_scaling_factor = compute_inv_freq(self.cfg, self.device, seq_len=needed)
            self._build_rope_table(self.cache.capacity)
            self._rope_shift_buf = None
            self._set_rope_shift(0)
            if self.cache.int4 and not self.fused_rope:
                raise ValueError("the int4 KV cache needs the attention kernel's own RoPE + append (full or interleaved "
                                 "rotary); this model rotates part of the head in rotate-half order")
        return self.cache

    def _build_rope_table(self, max_pos: int):
        """cos/sin table [max_pos, rot/2, 2] (fp32) read by the fused RoPE in the attention kernel; trig is
        evaluated once here in full precision instead of per token on the device."""
        cfg = self.cfg
        pos = torch.arange(max_pos, device=self.device, dtype=torch.float32)
        ang = pos[:, None] * self.inv_freq[None].float()
        self.no_rope = cfg.alibi or cfg.no_rope
        if self.no_rope:
            ang = torch.zeros_like(ang)          # identity rotation: the prompt path only appends k / v
        self.rope_table = torch.stack([ang.cos(), ang.sin()], -1).mul_(self.rope_scaling_factor).contiguous()
        self.fused_rope = self.no_rope or cfg.rope_interleaved or (cfg.rot_dim or cfg.head_dim) == cfg.head_dim

    def _set_rope_shift(self, off: int):
        """Decode position = cache length + ``off`` (non-zero after a compressed prompt)."""
        self._rope_shift = off
        if off == 0:
            self.rope_dec = self.rope_table
            return
        if self._rope_shift_buf is None:
            self._rope_shift_buf = torch.zeros_like(self.rope_table)
        n = self.rope_table.shape[0]
        if off > 0:                       # a compressed prompt: positions run ahead of the cache length
            self._rope_shift_buf[: n - off].copy_(self.rope_table[off:])
        else:                             # image tokens share positions: the cache is longer than the position count
            self._rope_shift_buf[-off:].copy_(self.rope_table[: n + off])
        self.rope_dec = self._rope_shift_buf

    @torch.no_grad()
    def evict_middle(self, start_size: int = 4, recent_size: int = 2000) -> int:
        """Attention-sink streaming (StreamingLLM, which the reference runs through the external ``streaming_llm``
        package: python/llm/example/GPU/Applications/streaming-llm/run_streaming_llama.py): keep the first
        ``start_size`` keys (the sinks) and the last ``recent_size``, drop what lies between, and let positions be the
        positions INSIDE the cache.  The cache holds keys that are already rotated, so the kept recent keys are rotated
        back by the number of dropped positions (rotations compose: R(p - drop) = R(-drop) R(p)); V^T columns only
        move.  fp16 cache only (re-quantizing an e5m2 / int4 key at every eviction would pile up rounding error).
        Returns the number of dropped positions.  Amortise it: evict down to ``recent_size`` whenever the next turn would
        not fit, not at every token."""
        cfg, cache = self.cfg, self.cache
        assert cache is not None and cache.slot_len is None, "evict_middle works on the plain (non-slot) cache"
        assert not cache.quantized, "attention-sink eviction needs the fp16 KV cache"
        assert self.fused_rope and not self._rope_shift and all(lw.mla is None for lw in self.layers), \
            "attention-sink eviction: plain rotary (or position-free) attention layers only"
        n = cache.host_len
        drop = n - start_size - recent_size
        if drop <= 0:
            return 0
        d = cfg.head_dim
        rot = 0 if self.no_rope else (cfg.rot_dim or d)
        if rot:
            ang = float(drop) * self.inv_freq.float().to(self.device)                 # [rot / 2]
            c, sn = ang.cos().to(torch.float32), ang.sin().to(torch.float32)
        for li in range(len(self.layers)):
            kc, vc = cache.k[li], cache.v[li]
            k = kc[:, :, start_size + drop:n].float()
            if rot:
                if cfg.rope_interleaved:
                    k1, k2 = k[..., 0:rot:2], k[..., 1:rot:2]
                    r1, r2 = k1 * c + k2 * sn, k2 * c - k1 * sn
                    k = torch.cat([torch.stack([r1, r2], -1).flatten(-2), k[..., rot:]], -1)
                else:
                    k1, k2 = k[..., : rot // 2], k[..., rot // 2: rot]
                    k = torch.cat([k1 * c + k2 * sn, k2 * c - k1 * sn, k[..., rot:]], -1)
            kc[:, :, start_size:start_size + recent_size] = k.to(kc.dtype)
            vc[:, :, :, start_size:start_size + recent_size] = vc[:, :, :, start_size + drop:n].clone()
        cache.past_len.fill_(start_size + recent_size)
        cache.host_len = start_size + recent_size
        return drop

    def _compress_plan(self, past, s, slot, nd):
        """The SnapKV settings that apply to this prompt pass, or None."""
        ck = self.compress_kv
        if not ck or past != 0 or slot is not None or nd or s <= ck.get("max_capacity_prompt", 1024):
            return None
        if self.cache.int4 or not self.fused_rope or self.no_rope or self.tp_group is not None:
            return None          # (int4 rows carry per-key scales in a second plane; ALiBi / learned positions count keys)
        if any(lw.window or lw.mla is not None for lw in self.layers):
            return None
        return dict(window_size=ck.get("window_size", 32), max_capacity_prompt=ck.get("max_capacity_prompt", 1024),
                    kernel_size=ck.get("kernel_size", 7), pooling=ck.get("pooling", "maxpool"))

    # ------------------------------------------------------------------ prefill (eager)
    def _linear_dense(self, x, lin: Lin):
        from ..ops import hip
        if lin.fused and x.shape[0] <= GEMV_MAX_ROWS:
            return hip.gemv_lowbit(x.contiguous(), lin.w, lin.qtype, lin.n, lin.k, bias=lin.bias)
        if lin.fused and x.shape[0] <= GEMM_MAX_ROWS:
            return hip.gemm_lowbit(x.contiguous(), lin.w, lin.qtype, lin.n, lin.k, bias=lin.bias)
        if lin.fused and self.prefill_fused and x.dtype == torch.float16 and x.dim() == 2 and not Lin.cache_dense \
                and not self.prefill_fp8:
            # prompt-sized rows: the fused prompt kernel (weights expanded once per workgroup into LDS; no fp16 copy of the
            # weight is written or read, no library GEMM)
            from ..transformers.low_bit_linear import lowbit_rows
            return lowbit_rows(x.contiguous(), lin.w, lin.qtype, lin.n, lin.k, bias=lin.bias)
        m = x.shape[0]
        if lin.fused and m > 4096 and x.dim() == 2 and not self.prefill_fp8:
            # hipBLASLt falls off a cliff right above 4096 rows (4100 - 4352 rows: 402 - 439 us against 272 at 4096 on the
            # qkv shape, 573 against 237 on down_proj) while 4096-row products run at its best rate: longer passes (packed
            # prompts + decode rows, long prompt chunks) go through it 4096 rows at a time, the rest as its own product
            # (at most 256 rows: the fused low-bit kernels).
            y = torch.empty(m, lin.n, dtype=x.dtype, device=x.device)
            w = self._dense_of(lin)
            m0 = m // 4096 * 4096
            for r in range(0, m0, 4096):
                torch.mm(x[r:r + 4096], w.t(), out=y[r:r + 4096])
            if lin.bias is not None:
                y[:m0] += lin.bias
            if m0 < m:
                y[m0:] = self._linear_dense(x[m0:], lin)
            return y
        if self.prefill_fp8 and x.dtype == torch.float16 and lin.k % 16 == 0 and lin.n % 16 == 0 and lin.k <= 16384:
            # opt-in: activations quantized per row to e4m3, weights per output row, hipBLASLt's fp8 GEMM (twice the fp16
            # MFMA rate on gfx950); lower precision than the fp16 product, never the default
            w8, sw = lin.fp8()
            x8, sx = hip.quant_rows_e4m3(x.contiguous())
            y = hip.bf16_to_f16(torch._scaled_mm(x8, w8.t(), scale_a=sx, scale_b=sw.t(), out_dtype=torch.bfloat16))
        else:
            y = F.linear(x, self._dense_of(lin))
        if lin.bias is not None:
            y = y + lin.bias
        return y

    def _dense_ahead(self, lin: Lin):
        """fp16 form of ``lin``'s weight for a library GEMM on the current stream, valid until the next-but-one call.
        Starts the expansion of the weight that follows ``lin`` in a prompt pass on the side stream."""
        from ..ops import hip
        a = self._ahead
        if a is None:
            plain = [lw for lw in self.layers if lw.mla is None and lw.moe is None]
            order = [l for lw in plain for l in (lw.qkv, lw.o, lw.gate, lw.up, lw.down) if l is not None and l.fused]
            a = self._ahead = dict(next={id(p): n for p, n in zip(order, order[1:])}, lin=[None, None], last=1,
                                   buf=[None, None], side=torch.cuda.Stream(device=self.device),
                                   ready=[torch.cuda.Event(), torch.cuda.Event()], free=[torch.cuda.Event(), torch.cuda.Event()])
            a["numel"] = max((l.n * l.k for l in order), default=0)
        main = torch.cuda.current_stream()

        def view(j, l):
            need = max(a["numel"], l.n * l.k)
            if a["buf"][j] is None or a["buf"][j].numel() < need:
                a["buf"][j] = torch.empty(need, dtype=torch.float16, device=self.device)
                a["buf"][j].record_stream(a["side"])
            return a["buf"][j][: l.n * l.k].view(l.n, l.k)

        j = 0 if a["lin"][0] is lin else 1 if a["lin"][1] is lin else None
        if j is not None:
            main.wait_event(a["ready"][j])
        else:
            j = 1 - a["last"]
            if a["lin"][j] is not None:
                main.wait_event(a["ready"][j])        # (a look-ahead nobody consumed may still be writing the buffer)
            w = view(j, lin)
            hip.dequant_t16(lin.w, lin.qtype, lin.n, lin.k, out=w)
            a["lin"][j] = lin
        a["last"] = j
        w = view(j, lin)
        nxt = a["next"].get(id(lin))
        if nxt is not None and a["lin"][1 - j] is not nxt:
            o = 1 - j
            a["free"][o].record(main)                 # the GEMM that read this buffer is already in the stream
            a["side"].wait_event(a["free"][o])
            with torch.cuda.stream(a["side"]):
                hip.dequant_t16(nxt.w, nxt.qtype, nxt.n, nxt.k, out=view(o, nxt))
                a["ready"][o].record(a["side"])
            a["lin"][o] = nxt
        return w

    def _dense_of(self, lin: Lin):
        if lin.fused and self.dequant_ahead and not Lin.cache_dense and not torch.cuda.is_current_stream_capturing():
            return self._dense_ahead(lin)
        return lin.dense()

    def _tp_reduce(self, t):
        if self.tp_group is not None:
            if self.tp_fast_allreduce is not None and self.tp_fast_allreduce.usable(t):
                self.tp_fast_allreduce(t)          # one kernel launch over IPC-mapped peer buffers (graph-capturable)
            elif t.is_cuda and torch.distributed.get_backend(self.tp_group) == "gloo":
                # host-staged sum, staged here: gloo's own device-tensor path (private stream + worker thread) was seen
                # to stall now and then with two ranks on one GPU; a blocking copy, a host all-reduce and a copy back
                # have nothing to race with
                h = t.cpu()
                torch.distributed.all_reduce(h, group=self.tp_group)
                t.copy_(h)
            else:
                torch.distributed.all_reduce(t, group=self.tp_group)
        return t

    def _reduce_capturable(self):
        """Whether the tensor-parallel sums of a decode step can sit inside a captured graph: the IPC kernels and RCCL's
        collectives can, a host-staged backend (gloo) cannot -- such a group runs its decode steps eagerly."""
        if self.tp_group is None or self.tp_fast_allreduce is not None:
            return True
        return torch.distributed.get_backend(self.tp_group) == "nccl"

    def _norm(self, x, w, b=None, residual=None, res_out=None):
        from ..ops import hip
        if self.cfg.norm == "layer":
            return hip.layer_norm(x, w, b, self.cfg.rms_eps)
        return hip.rms_norm(x, w, self.cfg.rms_eps, offset=self.cfg.norm_offset, residual=residual, res_out=res_out)

    def _add_norm(self, h, add, w, b=None):
        """(norm(h + add), h + add): the residual add rides in the RMSNorm launch (the prompt pass otherwise spends one
        elementwise launch per add: 64 of them, 0.37 ms, on a 1024-token Llama-2-7B prompt).  ``add`` None: (norm(h), h)."""
        from ..ops import hip
        if add is None:
            return self._norm(h, w, b), h
        if self.cfg.norm == "layer" or add.dtype != h.dtype or not (h.is_contiguous() and add.is_contiguous()):
            h = h + add
            return self._norm(h, w, b), h
        hn = torch.empty_like(h)
        x = hip.rms_norm(add, w, self.cfg.rms_eps, offset=self.cfg.norm_offset, residual=h, res_out=hn)
        return x, hn

    def _embed_extras(self, h, pos):
        """learned absolute positions and the embedding LayerNorm; h [rows, hidden], pos [rows] (device, int64)"""
        from ..ops import hip
        if self.pos_embed is not None:
            h = h + F.embedding(pos + self.cfg.pos_offset, self.pos_embed)
        if self.embed_norm is not None:
            h = hip.layer_norm(h, self.embed_norm[0], self.embed_norm[1], self.cfg.rms_eps)
        return h

    def _mlp_dense(self, x, lw):
        from ..ops import hip
        cfg = self.cfg
        if cfg.gated_mlp:
            gt, up = lw.gate, lw.up
            if (gt.fused and up.fused and gt.qtype == up.qtype and gt.n == up.n and gt.n % 16 == 0 and self.prefill_fused
                    and x.shape[0] > GEMM_MAX_ROWS and x.dtype == torch.float16 and not Lin.cache_dense and not self.prefill_fp8
                    and hip.gemm_prompt_supported(gt.qtype, gt.k)):
                # both input projections and act(g) * u in one launch: g and u never reach memory
                g = hip.gemm_prompt_pair(x.contiguous(), gt.w, up.w, gt.qtype, gt.n, gt.k, bias=gt.bias, bias2=up.bias,
                                         act=cfg.act)
            else:
                g = self._linear_dense(x, gt)
                u = self._linear_dense(x, up)
                hip.act_mul(g, u, cfg.act)
        else:
            g = _ACT_FNS[cfg.act](self._linear_dense(x, lw.up))
        return self._tp_reduce(self._linear_dense(g, lw.down))

    @torch.no_grad()
    def prefill(self, input_ids: Optional[torch.Tensor], last_only: bool = True, slot: Optional[int] = None,
                with_decode: bool = False, hidden: Optional[torch.Tensor] = None,
                inputs_embeds: Optional[torch.Tensor] = None, mrope_positions: Optional[torch.Tensor] = None):
        """input_ids [B, S] (device).  Fills the cache from position past_len and returns logits
        ([B, 1, V] if last_only else [B, S, V]).  With ``slot`` (continuous batching) input_ids is one sequence
        [1, S] that is appended to cache row ``slot`` only.

        ``with_decode`` (slot mode): the pass also carries one decode row per cache slot, the chained token of the
        running sequences, so that the prompt's GEMMs and the decode step share one read of the weights (what the
        reference's serving loop gets from vLLM's chunked prefill,
        python/llm/src/ipex_llm/vllm/xpu/engine/engine.py:40-246).  Linears, norms and the MLP run over the
        S + slots rows together; attention runs per kind (flash kernel for the prompt, split-KV decode kernel for the
        slots).  Returns (prompt logits, decode logits [slots, V], next tokens [slots]) and advances every active
        slot by one token exactly like ``decode(None, 1, chain=True)``."""
        from ..ops import hip
        cfg, cache = self.cfg, self.cache
        if not self.stage_first:
            assert hidden is not None and not with_decode, "a later pipeline stage is fed hidden states [B, S, hidden]"
            b, s = hidden.shape[:2]
        elif inputs_embeds is not None:
            # multimodal prompts: the caller has merged image features into the token embeddings
            b, s = inputs_embeds.shape[:2]
        else:
            b, s = input_ids.shape
        if slot is not None:
            assert b == 1 and cache.slot_len is not None, "slot prefill takes one sequence on a cache in slot mode"
            past = cache.slot_len[slot]
        else:
            past = cache.host_len
        assert past + s <= cache.capacity, "KV cache capacity exceeded"
        nd, st = 0, None
        if with_decode:
            assert slot is not None and last_only and not cache.active_host[slot], \
                "the mixed pass appends a new sequence to an idle slot"
            assert max(cache.slot_len) + 1 <= cache.capacity, "KV cache capacity exceeded"
            nd = cache.batch
            st = self._get_graph(nd, 1, True)[1]         # the decode step's static buffers: tokens, workspace
        if self.stage_first:
            if inputs_embeds is not None:
                assert not with_decode
                h = inputs_embeds.to(self.device, torch.float16)
            else:
                h = F.embedding(input_ids, self.embed).to(torch.float16)
            if cfg.embed_scale != 1.0:
                h = h * cfg.embed_scale
            h = h.reshape(b * s, cfg.hidden_size)
            pos = torch.arange(past, past + s, device=self.device)[None].expand(b, s)
            if nd:
                hd = F.embedding(st["tokens"].view(-1), self.embed).to(torch.float16)
                h = torch.cat([h, hd * cfg.embed_scale if cfg.embed_scale != 1.0 else hd])
                pos = torch.cat([pos.reshape(-1), cache.past_len.to(torch.int64)])
            h = self._embed_extras(h, pos.reshape(-1))
        else:
            h = hidden.to(torch.float16).reshape(b * s, cfg.hidden_size)
        hq, hkv, d = cfg.num_heads, cfg.num_kv_heads, cfg.head_dim
        snap = self._compress_plan(past, s, slot, nd)
        if past == 0 and slot is None and self._rope_shift:
            self._set_rope_shift(0)
        rope_table, rope_pos, mrope_delta = self.rope_table, None, 0
        if mrope_positions is not None:
            # one table row per prompt token: frequency i takes the angle of the token's position on the axis that
            # ``mrope_section`` assigns to i; the append / rotate kernel then indexes rows 0 .. B*S-1
            assert past == 0 and slot is None and not nd and cfg.mrope_section and self.fused_rope and not self.no_rope, \
                "3-axis rotary positions apply to a whole prompt on an engine built with mrope_section"
            mp = mrope_positions.to(self.device)
            if mp.shape[0] == 4:                       # (text, t, h, w): the text row duplicates what the engine counts
                mp = mp[1:]
            axis = torch.repeat_interleave(torch.arange(3, device=self.device),
                                           torch.tensor(list(cfg.mrope_section), device=self.device))
            assert axis.numel() == self.rope_table.shape[1], "mrope_section must cover rot_dim / 2 frequencies"
            sel = mp[axis].permute(1, 2, 0)                                        # [B, S, rot/2]
            rope_table = self.rope_table[sel, torch.arange(axis.numel(), device=self.device)].reshape(b * s, -1, 2).contiguous()
            rope_pos = torch.arange(b * s, device=self.device).view(b, s)
            top = mp.amax(dim=(0, 2))                                              # per sequence
            assert bool((top == top[0]).all()), "sequences of one batch must end at the same rotary position"
            mrope_delta = int(top[0]) + 1 - s
        pend = None                          # MLP output of the previous layer, added to h by the next norm launch
        for li, lw in enumerate(self.layers):
            x, h = self._add_norm(h, pend, lw.input_norm, lw.input_norm_b)
            pend = None
            qkv = self._linear_dense(x, lw.qkv) if lw.mla is None else self._mla_qkv_dense(x, lw.mla)
            if nd:
                # decode rows first: the idle slot's row appends a throw-away key at position 0 of the cache row
                # that the prompt's keys are written to right after
                attn_d = self._decode_attention(qkv[s:], li, lw, st, nd)
                qkv = qkv[:s]
            qkv = qkv.view(b, s, (hq + 2 * hkv) * d)
            q = qkv[..., : hq * d].view(b, s, hq, d).transpose(1, 2)
            k = qkv[..., hq * d: (hq + hkv) * d].view(b, s, hkv, d).transpose(1, 2)
            kc, vc = cache.k[li], cache.v[li]
            if slot is not None:
                kc, vc = kc[slot:slot + 1], vc[slot:slot + 1]
            # one launch: RoPE on q / k in place, K -> cache, V -> V^T cache (+ fp16 V^T scratch for e5m2 caches)
            rot = cfg.rot_dim or d
            scratch = None
            if cache.quantized and past == 0:
                scratch = self._vt_scratch(b, hkv, d, s)
            if cache.int4:
                # rotate in place, then quantize keys / values of the prompt into the int4 caches
                ksc, vsc = cache.k_scale[li], cache.v_scale[li]
                if slot is not None:
                    ksc, vsc = ksc[slot:slot + 1], vsc[slot:slot + 1]
                hip.prefill_rope_append(qkv, None, None, rope_table, n_heads=hq, n_kv_heads=hkv, head_dim=d,
                                        past=past, rot_dim=rot, interleaved=cfg.rope_interleaved, vt_scratch=scratch,
                                        position_ids=rope_pos)
                hip.kv_int4_append(qkv, kc, vc, ksc, vsc, n_heads=hq, n_kv_heads=hkv, head_dim=d, past=past)
            else:
                hip.prefill_rope_append(qkv, kc, vc, rope_table, n_heads=hq, n_kv_heads=hkv, head_dim=d, past=past,
                                        rot_dim=rot, interleaved=cfg.rope_interleaved, vt_scratch=scratch,
                                        position_ids=rope_pos)
            vt = scratch[..., :s] if scratch is not None else vc[:, :, :, :s]
            kv8 = False
            if past == 0:
                kk = k                       # attention reads the unquantized K / V^T of this call
            elif cache.fp8 and d in (64, 96, 128, 256):
                # a later chunk of the prompt: the kernel reads the e5m2 prefix (and this chunk's freshly appended
                # keys) straight from the cache
                rows_ = slice(None) if slot is None else slice(slot, slot + 1)
                kk, vt, kv8 = cache.k[li][rows_, :, :past + s], cache.v[li][rows_, :, :, :past + s], True
            else:
                kk, vt = self._read_cache(li, past + s, slot)
            attn = self._sdpa(q, kk, vt, past, lw.window, kv_e5m2=kv8).reshape(b * s, hq * d)
            if snap is not None:
                # the prompt has attended to all of its keys; what later tokens will see is the voted subset + the window
                from ..transformers.kv import snapkv_indices
                w_, cap_ = snap["window_size"], snap["max_capacity_prompt"]
                keep = snapkv_indices(q, k, num_key_value_groups=hq // hkv, scale=self.attn_scale, **snap)   # [b,hkv,cap-w]
                nk = cap_ - w_
                kc[:, :, :nk] = kc[:, :, : s - w_].gather(2, keep[..., None].expand(-1, -1, -1, d))
                kc[:, :, nk:cap_] = kc[:, :, s - w_: s].clone()
                vc[:, :, :, :nk] = vc[:, :, :, : s - w_].gather(3, keep[:, :, None, :].expand(-1, -1, d, -1))
                vc[:, :, :, nk:cap_] = vc[:, :, :, s - w_: s].clone()
            if nd:
                attn = torch.cat([attn, attn_d])
            o = self._tp_reduce(self._linear_dense(attn, lw.o))
            if lw.post_attn_out_norm is not None:
                o = self._norm(o, lw.post_attn_out_norm)
            if cfg.parallel_residual:
                x2 = x if lw.post_norm is None else self._norm(h, lw.post_norm, lw.post_norm_b)
                h = h + o + self._mlp_dense(x2, lw)
                continue
            x, h = self._add_norm(h, o, lw.pre_ffn_norm if lw.pre_ffn_norm is not None else lw.post_norm, lw.post_norm_b)
            if lw.moe is not None:
                m = lw.moe.forward_dense(x, self)
            else:
                m = self._mlp_dense(x, lw)
            if lw.post_ffn_out_norm is not None:
                m = self._norm(m, lw.post_ffn_out_norm)
            pend = m
        if pend is not None:
            h = h + pend
        if slot is not None:
            cache.past_len[slot] += s
            cache.slot_len[slot] += s
        elif snap is not None:
            cache.past_len.fill_(snap["max_capacity_prompt"])
            cache.host_len = snap["max_capacity_prompt"]
            self._set_rope_shift(s - snap["max_capacity_prompt"] + mrope_delta)
        else:
            cache.past_len.add_(s)
            cache.host_len += s
            if mrope_delta:
                self._set_rope_shift(mrope_delta)
        if nd:
            x = self._norm(h[s - 1:], self.final_norm, self.final_norm_b)            # last prompt token + the decode rows
            logits = self._finish_logits(self._linear_dense(x, self.lm_head).float())
            st["logits"].copy_(logits[1:])
            hip.argmax(st["logits"], out=st["tokens"].view(-1))                      # chained like the graph's step
            cache.past_len.add_(cache.active)
            cache.slot_len = [n + a for n, a in zip(cache.slot_len, cache.active_host)]
            return logits[:1].view(1, 1, cfg.vocab_size), st["logits"], st["tokens"].view(-1)
        h = h.view(b, s, -1)
        if not self.stage_last:
            return h                         # every position: the next stage needs them all for its own cache
        if last_only:
            h = h[:, -1:, :]
        x = self._norm(h.reshape(-1, cfg.hidden_size), self.final_norm, self.final_norm_b)
        logits = self._linear_dense(x, self.lm_head).float().view(b, -1, cfg.vocab_size)
        return self._finish_logits(logits)

    packed_with_decode = True            # prefill_packed(with_decode=True) exists (serving/continuous.py asks)

    @torch.no_grad()
    def prefill_packed(self, prompts, slots, with_decode: bool = False, hidden: Optional[torch.Tensor] = None,
                       lens: Optional[list] = None):
        """Several whole prompts (lists of token ids) into empty cache slots in ONE pass: norms, projections and the MLP
        run over the concatenated rows (one read of the weights for all prompts, what short chat prompts need: a
        32-token prompt alone costs 6 ms on a 7B model, mostly weight traffic), attention and the cache append run per
        prompt.  Returns the last-position logits [len(prompts), V].  The reference batches the prompt phase of
        several requests the same way (``max_prefilled_seqs``,
        python/llm/src/ipex_llm/transformers/pipeline_parallel.py:807-927).

        Pipeline stages: a later stage passes ``prompts=None`` with ``hidden`` [rows, hidden] (the previous stage's
        output for the same packed rows) and ``lens``; a stage that is not the last returns its [rows, hidden] output
        instead of logits."""
        from ..ops import hip
        cfg, cache = self.cfg, self.cache
        if not self.stage_first:
            assert hidden is not None and lens is not None and not with_decode, "a later stage is fed packed hidden rows"
            prompts = [None] * len(lens)
        assert cache.slot_len is not None and len(prompts) == len(slots) > 0
        nd, st = 0, None
        if with_decode:
            # one decode row per cache slot rides along, as in prefill(with_decode=True): returns
            # (prompt logits [n, V], decode logits [slots, V], next tokens [slots]) and advances every active slot
            assert max(cache.slot_len) + 1 <= cache.capacity, "KV cache capacity exceeded"
            nd = cache.batch
            st = self._get_graph(nd, 1, True)[1]
        lens = [len(p) for p in prompts] if self.stage_first else [int(n) for n in lens]
        assert all(cache.slot_len[sl] == 0 and not cache.active_host[sl] for sl in slots), "packed prompts go to empty slots"
        assert max(lens) <= cache.capacity, "KV cache capacity exceeded"
        offs = [0]
        for n in lens:
            offs.append(offs[-1] + n)
        if self.stage_first:
            ids = torch.tensor([t for p in prompts for t in p], dtype=torch.int64, device=self.device)
            h = F.embedding(ids, self.embed).to(torch.float16)
            if cfg.embed_scale != 1.0:
                h = h * cfg.embed_scale
            pos = torch.cat([torch.arange(n, device=self.device) for n in lens])
            if nd:
                hd = F.embedding(st["tokens"].view(-1), self.embed).to(torch.float16)
                h = torch.cat([h, hd * cfg.embed_scale if cfg.embed_scale != 1.0 else hd])
                pos = torch.cat([pos, cache.past_len.to(torch.int64)])
            h = self._embed_extras(h, pos)
        else:
            h = hidden.to(torch.float16).reshape(offs[-1], cfg.hidden_size)
        hq, hkv, d = cfg.num_heads, cfg.num_kv_heads, cfg.head_dim
        rot = cfg.rot_dim or d
        for li, lw in enumerate(self.layers):
            x = self._norm(h, lw.input_norm, lw.input_norm_b)
            qkv_all = self._linear_dense(x, lw.qkv) if lw.mla is None else self._mla_qkv_dense(x, lw.mla)
            parts = []
            attn_d = None
            if nd:
                # decode rows first: the rows of the slots being filled append a throw-away key at position 0 of cache rows
                # that the prompts' keys are written to right after
                attn_d = self._decode_attention(qkv_all[offs[-1]:], li, lw, st, nd)
            for j, slot in enumerate(slots):
                s = lens[j]
                qkv = qkv_all[offs[j]:offs[j + 1]].view(1, s, (hq + 2 * hkv) * d)
                kc, vc = cache.k[li][slot:slot + 1], cache.v[li][slot:slot + 1]
                scratch = self._vt_scratch(1, hkv, d, s) if cache.quantized else None
                if cache.int4:
                    hip.prefill_rope_append(qkv, None, None, self.rope_table, n_heads=hq, n_kv_heads=hkv, head_dim=d,
                                            past=0, rot_dim=rot, interleaved=cfg.rope_interleaved, vt_scratch=scratch)
                    hip.kv_int4_append(qkv, kc, vc, cache.k_scale[li][slot:slot + 1], cache.v_scale[li][slot:slot + 1],
                                       n_heads=hq, n_kv_heads=hkv, head_dim=d, past=0)
                else:
                    hip.prefill_rope_append(qkv, kc, vc, self.rope_table, n_heads=hq, n_kv_heads=hkv, head_dim=d, past=0,
                                            rot_dim=rot, interleaved=cfg.rope_interleaved, vt_scratch=scratch)
                q = qkv[..., : hq * d].view(1, s, hq, d).transpose(1, 2)
                k = qkv[..., hq * d: (hq + hkv) * d].view(1, s, hkv, d).transpose(1, 2)
                vt = scratch[..., :s] if scratch is not None else vc[:, :, :, :s]
                parts.append(self._sdpa(q, k, vt, 0, lw.window).reshape(s, hq * d))
            if attn_d is not None:
                parts.append(attn_d)
            o = self._tp_reduce(self._linear_dense(torch.cat(parts), lw.o))
            if lw.post_attn_out_norm is not None:
                o = self._norm(o, lw.post_attn_out_norm)
            if cfg.parallel_residual:
                x2 = x if lw.post_norm is None else self._norm(h, lw.post_norm, lw.post_norm_b)
                h = h + o + self._mlp_dense(x2, lw)
                continue
            h = h + o
            x = self._norm(h, lw.pre_ffn_norm if lw.pre_ffn_norm is not None else lw.post_norm, lw.post_norm_b)
            m = lw.moe.forward_dense(x, self) if lw.moe is not None else self._mlp_dense(x, lw)
            if lw.post_ffn_out_norm is not None:
                m = self._norm(m, lw.post_ffn_out_norm)
            h = h + m
        for j, slot in enumerate(slots):
            cache.past_len[slot] += lens[j]
            cache.slot_len[slot] += lens[j]
        if not self.stage_last:
            return h
        last = torch.tensor([o - 1 for o in offs[1:]] + list(range(offs[-1], offs[-1] + nd)), device=self.device)
        x = self._norm(h[last], self.final_norm, self.final_norm_b)
        logits = self._finish_logits(self._linear_dense(x, self.lm_head).float())
        if not nd:
            return logits
        n = len(slots)
        st["logits"].copy_(logits[n:])
        hip.argmax(st["logits"], out=st["tokens"].view(-1))                      # chained like the graph's step
        cache.past_len.add_(cache.active)
        cache.slot_len = [a + b for a, b in zip(cache.slot_len, cache.active_host)]
        return logits[:n], st["logits"], st["tokens"].view(-1)

    def _mla_qkv_dense(self, x, m: MLAWeights):
        """prompt rows: normed hidden [rows, hidden] -> qkv [rows, 3 * heads * head_dim] (un-rotated)"""
        from ..ops import hip
        cfg = self.cfg
        rows, qs, r = x.shape[0], cfg.q_size, cfg.rot_dim
        qkv = torch.empty(rows, 3 * qs, dtype=torch.float16, device=self.device)
        lat = self._linear_dense(x, m.a)
        ckv = lat[:, m.q_lora:]
        if m.q is not None:
            qkv[:, :qs] = self._linear_dense(x, m.q)
        else:
            qkv[:, :qs] = self._linear_dense(hip.rms_norm(lat[:, : m.q_lora].contiguous(), m.q_a_norm, cfg.rms_eps), m.q_b)
        kvn = hip.rms_norm(ckv[:, : m.kv_lora].contiguous(), m.kv_a_norm, cfg.rms_eps)
        qkv[:, qs:] = self._linear_dense(kvn, m.kv_b)
        # the rotary key is shared by all heads: it fills the first `rot_dim` channels of every K head
        qkv[:, qs:2 * qs].view(rows, cfg.num_heads, cfg.head_dim)[:, :, :r] = ckv[:, None, m.kv_lora:]
        return qkv

    def _mla_qkv_decode(self, h, lw, st):
        """the same inside the decode graph: GEMVs (input norm and the latent norms folded into them) + the copy
        of the shared rotary key (three GEMVs with a q latent: both latents come from one stacked weight)"""
        cfg, m = self.cfg, lw.mla
        qkv, lat, qs = st["qkv"], st["mla_lat"], cfg.q_size
        self._gemv(h, m.a, lat, gamma=lw.input_norm)
        ckv = lat[:, m.q_lora:]
        if m.q is not None:
            self._gemv(h, m.q, qkv[:, :qs], gamma=lw.input_norm)
        else:
            self._gemv(lat[:, : m.q_lora], m.q_b, qkv[:, :qs], gamma=m.q_a_norm)
        self._gemv(ckv[:, : m.kv_lora], m.kv_b, qkv[:, qs:], gamma=m.kv_a_norm)
        qkv[:, qs:2 * qs].view(-1, cfg.num_heads, cfg.head_dim)[:, :, : cfg.rot_dim].copy_(
            ckv[:, None, m.kv_lora:].expand(-1, cfg.num_heads, -1))

    def _decode_attention(self, qkv, li, lw, st, batch):
        """split-KV decode attention (+ RoPE and cache append) of one token per slot, outside the graph; qkv
        [batch, q + 2 kv] rows of a larger pass.  Returns [batch, q_size]."""
        from ..ops import hip
        cfg, cache = self.cfg, self.cache
        qkv = qkv.contiguous()
        out = torch.empty(batch, cfg.q_size, dtype=torch.float16, device=self.device)
        if not self.fused_rope:
            hip.rope_append(qkv, cache.k[li], cache.v[li], cache.past_len, self.inv_freq, n_heads=cfg.num_heads,
                            n_kv_heads=cfg.num_kv_heads, head_dim=cfg.head_dim, q_len=1, batch=batch,
                            rot_dim=cfg.rot_dim or cfg.head_dim, interleaved=cfg.rope_interleaved,
                            attn_scaling=self.rope_scaling_factor)
        hip.attn_decode(qkv, cache.k[li], cache.v[li], cache.past_len, out, st["ws"], st["counters"],
                        n_heads=cfg.num_heads, n_kv_heads=cfg.num_kv_heads, head_dim=cfg.head_dim, q_len=1,
                        batch=batch, scale=self.attn_scale, splits=st["splits"], softcap=cfg.attn_softcap,
                        window=lw.window, rope_table=self.rope_dec if self.fused_rope and not self.no_rope else None,
                        append=self.no_rope, alibi=self.alibi, interleaved=cfg.rope_interleaved, defer_combine=False,
                        **cache.scales(li))
        return out

    @torch.no_grad()
    def prefill_context_parallel(self, input_ids: torch.Tensor, group=None):
        """Prompt pass of ONE long sequence split over the ranks of ``group`` (parallel/context_parallel.py): this rank
        embeds and runs the linears for its two zigzag chunks only, K / V chunks are all-gathered into every rank's
        cache per layer, attention runs per chunk over the keys before it.  ``input_ids`` [1, S] is the full prompt on
        every rank, S a multiple of 2 * world; the cache must be empty.  Returns the last token's logits [1, 1, V] on
        every rank; afterwards every rank holds the complete KV cache and can decode."""
        from ..ops import hip
        from ..parallel import context_parallel as cp
        cfg, cache = self.cfg, self.cache
        assert not cache.int4, "context-parallel prefill gathers fp16 / e5m2 caches"
        b, s = input_ids.shape
        assert b == 1 and cache.host_len == 0 and cache.slot_len is None, "context-parallel prefill: one sequence, empty cache"
        assert s <= cache.capacity, "KV cache capacity exceeded"
        world, rank = cp._world(group), cp._rank(group)
        n = s // (2 * world)
        chunks = cp.zigzag_chunks(world, rank)
        pos = cp.local_token_index(s, world, rank, device=self.device)          # [2n]
        h = F.embedding(input_ids[0, pos], self.embed).to(torch.float16)
        if cfg.embed_scale != 1.0:
            h = h * cfg.embed_scale
        h = self._embed_extras(h, pos)
        hq, hkv, d = cfg.num_heads, cfg.num_kv_heads, cfg.head_dim
        rot = cfg.rot_dim or d
        for li, lw in enumerate(self.layers):
            x = self._norm(h, lw.input_norm, lw.input_norm_b)
            qkv = (self._linear_dense(x, lw.qkv) if lw.mla is None else self._mla_qkv_dense(x, lw.mla)).view(2, n, (hq + 2 * hkv) * d)
            kc, vc = cache.k[li], cache.v[li]
            for j, c in enumerate(chunks):      # RoPE at the chunk's global positions, K / V^T into the cache there
                hip.prefill_rope_append(qkv[j:j + 1], kc, vc, self.rope_table, n_heads=hq, n_kv_heads=hkv, head_dim=d,
                                        past=c * n, rot_dim=rot, interleaved=cfg.rope_interleaved)
            cp.gather_kv_chunks(kc, vc, n, group)
            outs = []
            for j, c in enumerate(chunks):
                q = qkv[j:j + 1, :, : hq * d].view(1, n, hq, d).transpose(1, 2)
                kk, vt = self._read_cache(li, (c + 1) * n)
                outs.append(self._sdpa(q, kk, vt, c * n, lw.window).reshape(n, hq * d))
            o = self._tp_reduce(self._linear_dense(torch.cat(outs, 0), lw.o))
            if lw.post_attn_out_norm is not None:
                o = self._norm(o, lw.post_attn_out_norm)
            if cfg.parallel_residual:
                x2 = x if lw.post_norm is None else self._norm(h, lw.post_norm, lw.post_norm_b)
                h = h + o + self._mlp_dense(x2, lw)
                continue
            h = h + o
            x = self._norm(h, lw.pre_ffn_norm if lw.pre_ffn_norm is not None else lw.post_norm, lw.post_norm_b)
            m = lw.moe.forward_dense(x, self) if lw.moe is not None else self._mlp_dense(x, lw)
            if lw.post_ffn_out_norm is not None:
                m = self._norm(m, lw.post_ffn_out_norm)
            h = h + m
        cache.past_len.fill_(s)
        cache.host_len = s
        # the last prompt token is the last row of rank 0's second chunk
        x = self._norm(h[-1:], self.final_norm, self.final_norm_b)
        logits = self._linear_dense(x, self.lm_head).float().view(1, 1, cfg.vocab_size)
        logits = self._finish_logits(logits)
        return cp.broadcast_from_last_token_owner(logits.contiguous(), group)

    def _finish_logits(self, logits):
        cfg = self.cfg
        if cfg.logit_scale != 1.0:
            logits = logits * cfg.logit_scale
        if cfg.final_softcap > 0:
            logits = torch.tanh(logits / cfg.final_softcap) * cfg.final_softcap
        return logits

    def _vt_scratch(self, b, hkv, d, s):
        need = (s + 7) // 8 * 8
        t = getattr(self, "_vt_buf", None)
        if t is None or t.shape[0] != b or t.shape[3] < need:
            self._vt_buf = t = torch.empty(b, hkv, d, need, dtype=torch.float16, device=self.device)
        return t

    def _read_cache(self, li, length, slot=None):
        """-> (K [B,Hkv,L,D], V^T [B,Hkv,D,L]) in fp16"""
        from ..ops import hip
        c = self.cache
        rows = slice(None) if slot is None else slice(slot, slot + 1)
        if c.int4:
            return hip.kv_int4_dequant(c.k[li][rows], c.v[li][rows], c.k_scale[li][rows], c.v_scale[li][rows], length)
        k = c.k[li][rows, :, :length]
        vt = c.v[li][rows, :, :, :length]
        if c.fp8:
            kf = torch.empty(k.shape, dtype=torch.float16, device=self.device)
            hip.from_e5m2(k, kf)
            vf = torch.empty(vt.shape, dtype=torch.float16, device=self.device)
            hip.from_e5m2(vt, vf)
            return kf, vf
        return k, vt

    def _sdpa(self, q, k, vt, past, window, kv_e5m2=False):
        """q [B,Hq,S,D], k [B,Hkv,past+S,D], vt = V^T [B,Hkv,D,past+S] -> [B, S, Hq*D]; with ``kv_e5m2`` k / vt are
        views of the fp8 cache (bytes)"""
        cfg = self.cfg
        if cfg.head_dim in (64, 96, 128, 256) and vt.stride(2) % 8 == 0 and vt.data_ptr() % 16 == 0:
            from ..ops import hip
            return hip.attn_prefill(q, k, vt, scale=self.attn_scale, causal=True, softcap=cfg.attn_softcap, window=window,
                                    v_transposed=True, alibi=self.alibi, kv_e5m2=kv_e5m2)
        assert not kv_e5m2
        if self.alibi is not None:
            raise NotImplementedError("ALiBi attention needs head_dim 64, 96 or 128")
        return self._sdpa_torch(q, k, vt.transpose(-1, -2), past, window).transpose(1, 2).reshape(q.shape[0], q.shape[2], -1)

    def _sdpa_torch(self, q, k, v, past, window):
        cfg = self.cfg
        s, L = q.shape[2], k.shape[2]
        simple = cfg.attn_softcap == 0 and not window and past == 0
        if simple:
            return F.scaled_dot_product_attention(q, k, v, is_causal=True, scale=self.attn_scale,
                                                  enable_gqa=cfg.num_heads != cfg.num_kv_heads)
        g = cfg.num_heads // cfg.num_kv_heads
        kk = k.repeat_interleave(g, 1) if g > 1 else k
        vv = v.repeat_interleave(g, 1) if g > 1 else v
        qpos = past + torch.arange(s, device=q.device)[:, None]
        kpos = torch.arange(L, device=q.device)[None]
        mask = kpos > qpos
        if window:
            mask |= kpos < qpos - window + 1
        if cfg.attn_softcap == 0:
            bias = torch.zeros(s, L, dtype=q.dtype, device=q.device).masked_fill(mask, float("-inf"))
            return F.scaled_dot_product_attention(q, kk, vv, attn_mask=bias, scale=self.attn_scale)
        sc = (q.float() @ kk.float().transpose(-1, -2)) * self.attn_scale
        sc = torch.tanh(sc / cfg.attn_softcap) * cfg.attn_softcap
        sc = sc.masked_fill(mask, float("-inf"))
        return (torch.softmax(sc, -1) @ vv.float()).to(q.dtype)

    # ------------------------------------------------------------------ decode (graph)
    def _build_static(self, batch: int, q_len: int):
        from ..ops import hip
        cfg = self.cfg
        m = batch * q_len
        dev = self.device
        splits = attn_decode_splits(batch, cfg.num_kv_heads, self.cache.capacity)
        # the residual stream is the payload of the message a pipeline stage hands to the next one; 8 spare halves behind
        # it carry the stop flag (bigdl_amd/parallel/pipeline.py)
        msg = torch.zeros(m * cfg.hidden_size + 8, dtype=torch.float16, device=dev)
        st = {
            "tokens": torch.zeros(batch, q_len, dtype=torch.int64, device=dev),
            "msg": msg,
            "h": msg[: m * cfg.hidden_size].view(m, cfg.hidden_size),
            "qkv": torch.zeros(m, cfg.q_size + 2 * cfg.kv_size, dtype=torch.float16, device=dev),
            "attn": torch.zeros(m, cfg.q_size, dtype=torch.float16, device=dev),
            "act": torch.zeros(m, max(cfg.intermediate_size, cfg.moe_intermediate_size, 8), dtype=torch.float16, device=dev),
            "tmp": torch.zeros(m, cfg.hidden_size, dtype=torch.float16, device=dev),
            "logits": torch.zeros(m, cfg.vocab_size, dtype=torch.float32, device=dev),
            "next": torch.zeros(m, dtype=torch.int64, device=dev),
            "ws": torch.empty(hip.attn_decode_ws_floats(batch, cfg.num_heads, cfg.num_kv_heads, q_len, cfg.head_dim, splits) + 16,
                              dtype=torch.float32, device=dev),
            "counters": torch.zeros(hip.attn_decode_counters(batch, cfg.num_heads, cfg.num_kv_heads), dtype=torch.int32,
                                    device=dev),
            "splits": splits,
        }
        if cfg.mla_kv_lora:
            st["mla_lat"] = torch.zeros(m, cfg.mla_q_lora + cfg.mla_kv_lora + cfg.rot_dim, dtype=torch.float16, device=dev)
        # with several KV splits the cross-split combine is folded into the o_proj GEMV's x staging
        g = cfg.num_heads // cfg.num_kv_heads
        rows = g * q_len
        kp = (cfg.q_size + 127) // 128 * 128
        st["defer"] = (splits > 1 and rows <= 32 and m <= 16 and m * (kp + 136) * 2 <= 48 * 1024
                       and not os.environ.get("BIGDL_NO_DEFER"))
        st["att_partials"] = (st["ws"], splits, cfg.head_dim, g, cfg.num_kv_heads, q_len, (rows + 15) // 16 * 16)
        return st

    def _gemv(self, x, lin: Lin, out, *, gamma=None, residual=None, w2: Lin = None, act=None, att_partials=None,
              beta=False):
        """out = epilogue(norm?(x) @ W^T): fused kernel for T16 weights, unfused torch ops otherwise.  ``beta``
        (a tensor or None, instead of the default False) selects LayerNorm(gamma, beta) as the norm."""
        from ..ops import hip
        cfg = self.cfg
        m = x.shape[0]
        if beta is not False:
            if lin.fused and (w2 is None or w2.fused) and m <= GEMV_MAX_ROWS and hip.gemv_x_in_lds(m, lin.k):
                return hip.gemv_lowbit_ln(x, lin.w, lin.qtype, lin.n, lin.k, gamma, lin.ln_fold(gamma, beta),
                                          w2=None if w2 is None else w2.w, ln2=None if w2 is None else w2.ln_fold(gamma, beta),
                                          bias=lin.bias, bias2=None if w2 is None else w2.bias, residual=residual,
                                          eps=cfg.rms_eps, act=act, out=out)
            x, gamma = hip.layer_norm(x, gamma, beta, cfg.rms_eps), None
        if (lin.fused and w2 is not None and w2.fused and m > GEMV_MAX_ROWS and not self.fold_batched_norm
                and w2.qtype == lin.qtype and w2.n == lin.n and not os.environ.get("BIGDL_NO_PAIR_GEMM")):
            # batched decode (17 .. 256 rows): gate and up in one launch, act(g) * u in its reduce kernel (three launches
            # fewer than two products + activation kernel: 10 us per layer at 64 rows)
            xin = x if gamma is None else hip.rms_norm(x, gamma, cfg.rms_eps, offset=cfg.norm_offset)
            return hip.gemm_lowbit_ex(xin, lin.w, lin.qtype, lin.n, lin.k, w2=w2.w, bias=lin.bias, bias2=w2.bias, act=act, out=out)
        if (lin.fused and (w2 is None or w2.fused) and m > GEMV_MAX_ROWS and att_partials is None and self.fold_batched_norm
                and (w2 is None or (w2.qtype == lin.qtype and w2.n == lin.n))):
            # BIGDL_FOLD_NORM=1 (measured slower, kept for experiments): the norm folded into the product (gamma while x
            # is staged: +7 us per product at 64 rows, more than the norm kernel it saves; 1 / rms from the row sums of
            # squares that the producer of x left behind), and a product that adds to the residual stream leaves the sums
            # of squares for the next norm
            ssq = None if gamma is None else self._row_ssq_of(x)
            ssq_out = self._row_ssq_for(out) if residual is not None and out.shape[1] == cfg.hidden_size else None
            return hip.gemm_lowbit_ex(x, lin.w, lin.qtype, lin.n, lin.k, w2=None if w2 is None else w2.w, bias=lin.bias,
                                      bias2=None if w2 is None else w2.bias, residual=residual, gamma=gamma,
                                      gamma_offset=cfg.norm_offset, eps=cfg.rms_eps, row_ssq=ssq, ssq_out=ssq_out,
                                      act=act, out=out)
        if lin.fused and (w2 is None or w2.fused) and m > GEMV_MAX_ROWS and att_partials is None:
            # batched decode (17 .. 64 rows): RMSNorm kernel + MFMA GEMM with the epilogue fused
            xin = x if gamma is None else hip.rms_norm(x, gamma, cfg.rms_eps, offset=cfg.norm_offset)
            if w2 is None:
                return hip.gemm_lowbit(xin, lin.w, lin.qtype, lin.n, lin.k, bias=lin.bias, residual=residual, act=act, out=out)
            g = hip.gemm_lowbit(xin, lin.w, lin.qtype, lin.n, lin.k, bias=lin.bias, out=out)
            u = hip.gemm_lowbit(xin, w2.w, w2.qtype, w2.n, w2.k, bias=w2.bias)
            return hip.act_mul(g, u, act)
        if lin.fused and (w2 is None or w2.fused):
            return hip.gemv_lowbit(x, lin.w, lin.qtype, lin.n, lin.k, w2=None if w2 is None else w2.w, bias=lin.bias,
                                   bias2=None if w2 is None else w2.bias, residual=residual, gamma=gamma,
                                   eps=cfg.rms_eps, gamma_offset=cfg.norm_offset, act=act, out=out,
                                   att_partials=att_partials)
        if (not lin.fused and (w2 is None or not w2.fused) and att_partials is None and hip.gemv_f16_ok(m, lin.n, lin.k)
                and lin.w.is_contiguous() and (w2 is None or w2.w.is_contiguous())):
            # plain fp16 weights: one kernel per projection, streamed row-major as stored
            return hip.gemv_f16(x, lin.w, w2=None if w2 is None else w2.w, bias=lin.bias,
                                bias2=None if w2 is None else w2.bias, residual=residual, gamma=gamma, eps=cfg.rms_eps,
                                gamma_offset=cfg.norm_offset, act=act, out=out)
        xin = x if gamma is None else hip.rms_norm(x, gamma, cfg.rms_eps, offset=cfg.norm_offset)
        y = F.linear(xin, lin.dense())
        if lin.bias is not None:
            y = y + lin.bias
        if w2 is not None:
            u = F.linear(xin, w2.dense())
            if w2.bias is not None:
                u = u + w2.bias
            y = hip.act_mul(y, u, act)
        if residual is not None:
            y = y + residual
        out.copy_(y)
        return out

    # ---- row sums of squares handed from the product that wrote a tensor to the folded norm of the next product ----
    def _ssq_reset(self, rows: int):
        """start of a decode step: zero the ring of [rows] fp32 accumulators and forget who owns what"""
        ring = self._ssq_ring.get(rows)
        if ring is None:
            ring = self._ssq_ring[rows] = torch.zeros(6 * len(self.layers) + 8, rows, dtype=torch.float32, device=self.device)
        else:
            ring.zero_()
        self._ssq_next = 0
        self._ssq_owner = None          # (data_ptr, width of the tensor the current sums belong to, buffer)

    def _ssq_slot(self, rows: int):
        ring = self._ssq_ring[rows]
        assert self._ssq_next < ring.shape[0], "row-sum ring exhausted within one decode step"
        slot = ring[self._ssq_next]             # every slot is used once per step: it has to start from zero
        self._ssq_next += 1
        return slot

    def _row_ssq_for(self, out):
        """accumulator for the rows about to be written to ``out`` (the product's reduce kernel fills it)"""
        slot = self._ssq_slot(out.shape[0])
        self._ssq_owner = (out.data_ptr(), out.shape[1], slot)
        return slot

    def _row_ssq_of(self, x):
        """sums of squares of the rows of x: left behind by the product that wrote x, else computed here"""
        if self._ssq_owner is not None and self._ssq_owner[:2] == (x.data_ptr(), x.shape[1]):
            return self._ssq_owner[2]
        slot = self._ssq_slot(x.shape[0])
        torch.sum(x.float().square(), dim=1, out=slot)
        self._ssq_owner = (x.data_ptr(), x.shape[1], slot)
        return slot

    def _mlp_in(self, h, lw, a, gamma, beta):
        """a = act(gate(norm(h))) * up(norm(h))  or  act(up(norm(h)))"""
        cfg = self.cfg
        if cfg.gated_mlp:
            self._gemv(h, lw.gate, a, gamma=gamma, beta=beta, w2=lw.up, act=cfg.act)
        else:
            self._gemv(h, lw.up, a, gamma=gamma, beta=beta, act=cfg.act)

    def _decode_body(self, st, batch: int, q_len: int, chain: bool):
        from ..ops import hip
        cfg, cache = self.cfg, self.cache
        h, qkv, attn, act, tmp = st["h"], st["qkv"], st["attn"], st["act"], st["tmp"]
        self.fold_batched_norm = (batch * q_len > GEMV_MAX_ROWS and cfg.norm != "layer" and self.tp_group is None
                                  and os.environ.get("BIGDL_FOLD_NORM") == "1")
        if self.fold_batched_norm:
            self._ssq_reset(batch * q_len)
        if not self.stage_first:
            pass                              # st["h"] holds the previous stage's output
        elif cfg.embed_scale != 1.0 or self.embed.dtype != h.dtype:
            h.copy_(F.embedding(st["tokens"].view(-1), self.embed) * cfg.embed_scale)
        else:
            torch.index_select(self.embed, 0, st["tokens"].view(-1), out=h)      # gather straight into the residual stream
        layer_norm = cfg.norm == "layer"
        if self.stage_first and (self.pos_embed is not None or self.embed_norm is not None):
            pos = cache.past_len.to(torch.int64)[:, None] + torch.arange(q_len, device=self.device)[None]
            h.copy_(self._embed_extras(h, pos.reshape(-1)))
        for li_, lw in enumerate(self.layers):
            # a layer-skipping draft engine (spec_engine.layer_skip_draft_of) runs a subset of the target's layers on
            # the target's cache: its layer li_ reads and writes the cache rows of the layer it was taken from
            li = li_ if self.cache_layer is None else self.cache_layer[li_]
            if lw.mla is not None:
                self._mla_qkv_decode(h, lw, st)
            elif layer_norm:
                self._gemv(h, lw.qkv, qkv, gamma=lw.input_norm, beta=lw.input_norm_b)
            else:
                self._gemv(h, lw.qkv, qkv, gamma=lw.input_norm)
            if not self.fused_rope:
                hip.rope_append(qkv, cache.k[li], cache.v[li], cache.past_len, self.inv_freq, n_heads=cfg.num_heads,
                                n_kv_heads=cfg.num_kv_heads, head_dim=cfg.head_dim, q_len=q_len, batch=batch,
                                rot_dim=cfg.rot_dim or cfg.head_dim, interleaved=cfg.rope_interleaved,
                                attn_scaling=self.rope_scaling_factor)
            hip.attn_decode(qkv, cache.k[li], cache.v[li], cache.past_len, attn, st["ws"], st["counters"],
                            n_heads=cfg.num_heads, n_kv_heads=cfg.num_kv_heads, head_dim=cfg.head_dim, q_len=q_len,
                            batch=batch, scale=self.attn_scale, splits=st["splits"], softcap=cfg.attn_softcap,
                            window=lw.window,
                            rope_table=self.rope_dec if self.fused_rope and not self.no_rope else None,
                            append=self.no_rope, alibi=self.alibi,
                            interleaved=cfg.rope_interleaved, defer_combine=st["defer"] and lw.o.fused, **cache.scales(li))
            att_partials = st["att_partials"] if st["defer"] and lw.o.fused else None
            if layer_norm and cfg.parallel_residual:
                # the MLP reads the layer input: its first GEMV runs before o_proj adds the attention output to h
                a = act[:, : lw.up.n]
                if lw.post_norm is None:
                    self._mlp_in(h, lw, a, lw.input_norm, lw.input_norm_b)
                else:
                    self._mlp_in(h, lw, a, lw.post_norm, lw.post_norm_b)
            if lw.post_attn_out_norm is None and self.tp_group is None:
                self._gemv(attn, lw.o, h, residual=h, att_partials=att_partials)
            else:
                self._tp_reduce(self._gemv(attn, lw.o, tmp, att_partials=att_partials))
                if lw.post_attn_out_norm is not None:
                    hip.rms_norm(tmp, lw.post_attn_out_norm, cfg.rms_eps, offset=cfg.norm_offset, out=tmp)
                h.add_(tmp)
                self._ssq_owner = None
            ffn_norm = lw.pre_ffn_norm if lw.pre_ffn_norm is not None else lw.post_norm
            if lw.moe is not None:
                lw.moe.forward_decode(self, h, ffn_norm)
                self._ssq_owner = None
                continue
            a = act[:, : lw.up.n]
            if not (layer_norm and cfg.parallel_residual):
                self._mlp_in(h, lw, a, ffn_norm, lw.post_norm_b if layer_norm else False)
            if lw.post_ffn_out_norm is None and self.tp_group is None:
                self._gemv(a, lw.down, h, residual=h)
            else:
                self._tp_reduce(self._gemv(a, lw.down, tmp))
                if lw.post_ffn_out_norm is not None:
                    hip.rms_norm(tmp, lw.post_ffn_out_norm, cfg.rms_eps, offset=cfg.norm_offset, out=tmp)
                h.add_(tmp)
                self._ssq_owner = None
        if not self.stage_last:
            cache.past_len.add_(cache.active, alpha=q_len)
            return
        self._gemv(h, self.lm_head, st["logits"], gamma=self.final_norm, beta=self.final_norm_b if layer_norm else False)
        if cfg.logit_scale != 1.0 or cfg.final_softcap > 0:
            st["logits"].copy_(self._finish_logits(st["logits"]))
        # the argmax launch also advances the cache lengths (past_len += q_len * active)
        if chain and q_len == 1:
            # the greedy token is written where the next replay reads its input; "next" aliases that buffer
            hip.argmax(st["logits"], out=st["tokens"].view(-1), inc=(cache.past_len, cache.active, q_len))
        else:
            hip.argmax(st["logits"], out=st["next"], inc=(cache.past_len, cache.active, q_len))

    def _get_graph(self, batch: int, q_len: int, chain: bool):
        key = (batch, q_len, chain, self._rope_shift != 0)      # (the shifted rotary table is a different buffer)
        if key in self._graphs:
            return self._graphs[key], self._static[(batch, q_len)]
        st = self._static.get((batch, q_len))
        if st is None:
            st = self._static[(batch, q_len)] = self._build_static(batch, q_len)
        if not self.use_graph or not self._reduce_capturable():
            self._graphs[key] = None
            return None, st
        # warm up on a side stream (lazy module loads must happen outside capture), then capture.
        # The warm-up step really appends to the cache, so snapshot and restore the length counter.
        saved_len = self.cache.past_len.clone()
        saved_tok = st["tokens"].clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._decode_body(st, batch, q_len, chain)
            self.cache.past_len.copy_(saved_len)
            st["tokens"].copy_(saved_tok)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                self._decode_body(st, batch, q_len, chain)
        torch.cuda.current_stream().wait_stream(side)
        self._graphs[key] = g
        return g, st

    @torch.no_grad()
    def decode(self, tokens: Optional[torch.Tensor], q_len: int = 1, chain: bool = False):
        """One decode step for `tokens` [B, q_len] (None: reuse the chained token already on the device).
        Returns (logits [B*q_len, V] fp32 view of a static buffer, next_tokens [B*q_len])."""
        batch = self.cache.batch
        c = self.cache
        if c.slot_len is not None:
            assert max(c.slot_len) + q_len <= c.capacity, "KV cache capacity exceeded"
        else:
            assert c.host_len + q_len <= c.capacity, "KV cache capacity exceeded"
        g, st = self._get_graph(batch, q_len, chain)
        if tokens is not None:
            st["tokens"].copy_(tokens.view(batch, q_len))
        if g is None:
            self._decode_body(st, batch, q_len, chain)
        else:
            g.replay()
        if c.slot_len is not None:
            c.slot_len = [n + q_len * a for n, a in zip(c.slot_len, c.active_host)]
        else:
            c.host_len += q_len
        return st["logits"], (st["tokens"].view(-1) if chain and q_len == 1 else st["next"])

    # ------------------------------------------------------------------ generation
    # prompt lengths up to 128 are padded to one of these and replayed as a captured pass (Llama-2-7B first token, eager ->
    # captured: 32 tokens 5.48 -> 3.65 ms, 128 tokens 5.73 -> 5.61 ms; from 256 tokens on the kernels outlast their launches)
    PREFILL_BUCKETS = (16, 32, 48, 64, 96, 128)

    def _prefill_short_ok(self, b, s, inputs_embeds, mrope_positions):
        """whether a fresh prompt of this shape takes the captured pass of ``_prefill_short``"""
        return (self.use_graph and b == 1 and s <= self.PREFILL_BUCKETS[-1] and inputs_embeds is None
                and mrope_positions is None and self.stage_first and self.stage_last and self.tp_group is None
                and not self.compress_kv and not getattr(self, "prefill_fp8", False)
                and self.cache.host_len == 0 and self.cache.slot_len is None and not self.cache.int4
                and self.PREFILL_BUCKETS[-1] <= self.cache.capacity     # (pad rows are written up to the bucket's length)
                and os.environ.get("BIGDL_PREFILL_GRAPH", "1") != "0"
                and not any(lw.moe is not None or lw.mla is not None for lw in self.layers))

    @torch.no_grad()
    def _prefill_short(self, input_ids):
        """Logits [1, V] of the last prompt token for a fresh prompt of at most 128 tokens, with the whole prompt pass
        replayed as ONE hipGraph.  A 32-token prompt pass is about 400 kernels of a few microseconds each: issued one by
        one the host needs 5.6 ms for them on a Llama-2-7B, more than three decode steps.  The prompt is padded to a
        length bucket (pad rows sit behind the real ones, so no real token attends to them), the pass over the bucket
        is captured once per bucket and cache, and the cache length is set back to the real length afterwards (the pad
        rows' keys are overwritten by the tokens that follow)."""
        s = input_ids.shape[1]
        bucket = next(bk for bk in self.PREFILL_BUCKETS if bk >= s)
        ent = self._prefill_graphs.get(bucket)
        if ent is None:
            ids_buf = torch.zeros(1, bucket, dtype=torch.int64, device=self.device)
            ids_buf[:, :s].copy_(input_ids)
            self.prefill(ids_buf, last_only=False)            # eager once: workspaces and kernel attributes settle
            torch.cuda.synchronize()
            self.cache.reset()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                logits = self.prefill(ids_buf, last_only=False)
            ent = self._prefill_graphs[bucket] = (graph, ids_buf, logits)
            self.cache.reset()
        graph, ids_buf, logits = ent
        if s < bucket:
            ids_buf[:, s:].zero_()
        ids_buf[:, :s].copy_(input_ids)
        graph.replay()
        self.cache.crop(s)
        return logits[:, s - 1, :]

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, max_new_tokens: int = 32, min_new_tokens: int = 0, do_sample=False,
                 temperature=1.0, top_k=0, top_p=1.0, repetition_penalty=1.0, eos_token_id=None, pad_token_id=None,
                 streamer=None, kv_fp8: bool = False, stopping=None, inputs_embeds=None, mrope_positions=None,
                 attention_mask=None):
        """Greedy / sampling generation with per-token timing in the reference benchmark's terms.  ``inputs_embeds``
        [B, S, hidden] / ``mrope_positions`` [3, B, S]: a multimodal prompt as its wrapper model prepared it (the
        token ids are still what is returned in front of the generated tokens).  ``attention_mask`` [B, S] with
        padding (left or right): every row becomes a cache slot with its own length -- the prompts run packed in one
        pass without their pad tokens, positions count from each row's first real token, and the decode steps read
        the per-slot lengths on the device."""
        from ..ops import hip
        dev = self.device
        input_ids = input_ids.to(dev)
        b, s = input_ids.shape
        ragged = attention_mask is not None and not bool(attention_mask.all())
        self.ensure_cache(b, s + max_new_tokens + 1, fp8=kv_fp8)
        self.cache.reset()
        eos = set([eos_token_id] if isinstance(eos_token_id, int) else (eos_token_id or []))
        greedy = not do_sample and repetition_penalty == 1.0
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        if ragged:
            assert inputs_embeds is None and mrope_positions is None, "padded batches carry token ids only"
            keep = attention_mask.to(dev).bool()
            prompts = [input_ids[i][keep[i]].tolist() for i in range(b)]
            assert all(prompts), "a batch row without any real token"
            self.cache.enable_slots()
            logits = self.prefill_packed(prompts, list(range(b)))
            for i in range(b):
                self.cache.set_active(i, True)
        else:
            if self._prefill_short_ok(b, s, inputs_embeds, mrope_positions):
                logits = self._prefill_short(input_ids)
            else:
                logits = self.prefill(input_ids, inputs_embeds=inputs_embeds, mrope_positions=mrope_positions)[:, -1, :]
        out = [input_ids]
        unfinished = torch.ones(b, dtype=torch.bool, device=dev)
        all_ids = input_ids
        times = []
        nxt = None
        if streamer is not None:
            streamer.put(input_ids.cpu())       # the prompt first, as HF generate does (skip_prompt drops exactly this)
        # (Queuing step k + 1 before the host waits for token k -- tokens handed to the host through a side stream and
        # events -- was measured and dropped: 1.697 against 1.563 ms per token.  The host's turnaround is only 12 us of the
        # step, and the cross-stream hand-off slowed every kernel of the graph that ran beside it.)
        for step in range(max_new_tokens):
            if step > 0:
                if greedy:
                    logits_flat, nxt_dev = self.decode(None if step > 1 else nxt.view(b, 1), 1, chain=True)
                    nxt = nxt_dev.clone()
                else:
                    logits_flat, _ = self.decode(nxt.view(b, 1), 1, chain=False)
                    logits = logits_flat.view(b, -1)
            if step == 0 or not greedy:
                lg = logits.float()
                if repetition_penalty != 1.0:
                    lg = lg.contiguous()
                    hip.repetition_penalty_inplace(lg, all_ids, repetition_penalty)
                if do_sample:
                    nxt = sample_logits(lg, temperature, top_k, top_p)
                else:
                    nxt = lg.argmax(-1)
            if eos and step < min_new_tokens:
                pass
            if pad_token_id is not None:
                nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad_token_id))
            out.append(nxt.view(b, 1))
            if not greedy or repetition_penalty != 1.0:
                all_ids = torch.cat([all_ids, nxt.view(b, 1)], 1)
            if streamer is not None:
                streamer.put(nxt.view(b, 1).cpu() if b > 1 else nxt.cpu())
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            times.append(t1 - t0)
            t0 = t1
            if eos and step + 1 >= min_new_tokens:
                host = nxt.tolist()
                for i, t in enumerate(host):
                    if t in eos:
                        unfinished[i] = False
                if not bool(unfinished.any()):
                    break
            if stopping is not None and stopping(torch.cat(out, 1)):
                break
        if streamer is not None:
            streamer.end()
        if self.tp_fast_allreduce is not None:
            # the IPC all-reduce kernels give up on a peer after a bounded spin and flag it: never return tokens that
            # were computed from a partial sum
            self.tp_fast_allreduce.check()
        self.first_cost = self.first_token_time = times[0]
        self.last_token_time = times[1:]
        self.rest_cost_mean = sum(times[1:]) / max(1, len(times) - 1)
        self.peak_memory = torch.cuda.max_memory_reserved() / (1024 ** 3)
        return torch.cat(out, 1)

    def dense_copy_bytes(self):
        """HBM an fp16 copy of every fused linear takes (what ``Lin.cache_dense`` keeps resident for the prompt GEMMs)"""
        lins = [self.lm_head] * (self.lm_head is not None) + [lin for lw in self.layers for lin in (lw.qkv, lw.o, lw.gate, lw.up, lw.down)]
        lins += [lin for lw in self.layers if lw.mla is not None for lin in lw.mla.lins()]
        return sum(2 * lin.n * lin.k for lin in lins if lin is not None and lin.fused)

    def weight_bytes(self):
        tot = 0 if self.embed is None else self.embed.numel() * self.embed.element_size()
        tot += 0 if self.lm_head is None else self.lm_head.nbytes
        for lw in self.layers:
            for lin in (lw.qkv, lw.o, lw.gate, lw.up, lw.down):
                if lin is not None:
                    tot += lin.nbytes
            if lw.moe is not None:
                tot += lw.moe.nbytes
            if lw.mla is not None:
                tot += sum(lin.nbytes for lin in lw.mla.lins())
        return tot

    def decode_read_bytes(self):
        """Weight bytes one decoded token has to stream from HBM: every dense product once, the routed experts a token
        activates (not all of them), the head; the embedding contributes one row.  The denominator of a bandwidth
        floor for batch-1 decode."""
        tot = 0 if self.lm_head is None else self.lm_head.nbytes
        if self.embed is not None:
            tot += self.embed.shape[-1] * self.embed.element_size()
        for lw in self.layers:
            for lin in (lw.qkv, lw.o, lw.gate, lw.up, lw.down):
                if lin is not None:
                    tot += lin.nbytes
            if lw.moe is not None:
                tot += lw.moe.active_bytes()
            if lw.mla is not None:
                tot += sum(lin.nbytes for lin in lw.mla.lins())
        return tot


def sample_logits(logits, temperature=1.0, top_k=0, top_p=1.0):
    if temperature and temperature != 1.0:
        logits = logits / temperature
    if top_k and top_k > 0:
        kth = torch.topk(logits, min(top_k, logits.shape[-1]), dim=-1).values[..., -1:]
        logits = logits.masked_fill(logits < kth, float("-inf"))
    if top_p < 1.0:
        srt, idx = torch.sort(logits, descending=True, dim=-1)
        cum = torch.softmax(srt, -1).cumsum(-1)
        drop = cum - torch.softmax(srt, -1) > top_p
        srt = srt.masked_fill(drop, float("-inf"))
        logits = torch.full_like(logits, float("-inf")).scatter(-1, idx, srt)
    return torch.multinomial(torch.softmax(logits, -1), 1).squeeze(-1)


# ---------------------------------------------------------------------------------------------
# synthetic models (benchmarks / smoke tests: random-init quantized weights of a named architecture)
# ---------------------------------------------------------------------------------------------
def _random_t16(n, k, qtype, device, gen, scale=0.02):
    """A random quantized [n, k] weight created directly on the device (no 28 GB fp32 detour):
    uniform random codes, block scales ~ U(0.5, 1.5) * scale / 8."""
    from ..ops import host
    nb = host.device_nbytes(qtype, n, k)
    np_, kp = (n + 15) // 16 * 16, (k + 127) // 128 * 128
    lo_bits = 8 if qtype in (Q.SYM_INT8, Q.FP8E4, Q.FP8E5) else 4
    hi_bits = {Q.SYM_INT5: 1, Q.ASYM_INT5: 1, Q.FP6: 2, Q.Q5_K: 1}.get(qtype, 0)
    code_bytes = np_ * kp * (lo_bits + hi_bits) // 8
    buf = torch.empty(nb, dtype=torch.uint8, device=device)
    codes = torch.randint(0, 256, (code_bytes,), dtype=torch.uint8, device=device, generator=gen)
    if qtype in (Q.FP8E4, Q.FP8E5):
        codes &= 0xBF  # keep the exponent's top bit clear: no inf/nan, magnitudes <= 2
    buf[:code_bytes] = codes
    nscale = np_ * (kp // 32)
    div = {Q.SYM_INT8: 127.0, Q.FP8E4: 1.0, Q.FP8E5: 1.0, Q.NF4: 1.0, Q.FP4: 6.0, Q.FP6: 28.0}.get(qtype, 8.0)
    sc = ((torch.rand(nscale, device=device, generator=gen) + 0.5) * (scale / div)).to(torch.float16)
    buf[code_bytes:code_bytes + nscale * 2] = sc.view(torch.uint8)
    if nb > code_bytes + nscale * 2:  # mins: centre asymmetric formats around zero
        mn = (-sc.float() * (2 ** (lo_bits + hi_bits) - 1) / 2).to(torch.float16)
        buf[code_bytes + nscale * 2:] = mn.view(torch.uint8)
    p = FP4Params(buf, quantized=True, _shape=(n, k), qtype=qtype, qlayout="t16")
    return p


@torch.no_grad()
def dense_reference_logits(model: "NativeDecoder", ids: torch.Tensor) -> torch.Tensor:
    """Logits [B, S, V] of a plain fp32 PyTorch forward over the dequantized weights of a llama-shaped engine (RMSNorm,
    half-split rotary, gated MLP, no biases): the oracle of the engine tests and of ``__graft_entry__.smoke``."""
    cfg = model.cfg
    dev = ids.device
    h = model.embed[ids].float() * cfg.embed_scale
    b, s = ids.shape
    hq, hkv, d = cfg.num_heads, cfg.num_kv_heads, cfg.head_dim
    ang = torch.arange(s, device=dev).float()[:, None] * model.inv_freq.float()[None]
    cos, sin = ang.cos(), ang.sin()

    def norm(x, w):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + cfg.rms_eps) * (w.float() + cfg.norm_offset)

    def rope(x):
        x0, x1 = x[..., : d // 2], x[..., d // 2:]
        return torch.cat([x0 * cos - x1 * sin, x1 * cos + x0 * sin], -1)

    future = torch.triu(torch.ones(s, s, dtype=torch.bool, device=dev), 1)
    for lw in model.layers:
        x = norm(h, lw.input_norm)
        qkv = x @ lw.qkv.dense(torch.float32).float().t()
        q = rope(qkv[..., : hq * d].view(b, s, hq, d).transpose(1, 2))
        k = rope(qkv[..., hq * d:(hq + hkv) * d].view(b, s, hkv, d).transpose(1, 2)).repeat_interleave(hq // hkv, 1)
        v = qkv[..., (hq + hkv) * d:].view(b, s, hkv, d).transpose(1, 2).repeat_interleave(hq // hkv, 1)
        sc = (q @ k.transpose(-1, -2) / math.sqrt(d)).masked_fill(future, float("-inf"))
        a = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(b, s, hq * d)
        h = h + a @ lw.o.dense(torch.float32).float().t()
        x = norm(h, lw.post_norm)
        g = F.silu(x @ lw.gate.dense(torch.float32).float().t()) * (x @ lw.up.dense(torch.float32).float().t())
        h = h + g @ lw.down.dense(torch.float32).float().t()
    return norm(h, model.final_norm) @ model.lm_head.dense(torch.float32).float().t()


def build_random_decoder(cfg: DecoderConfig, qtype: int = Q.SYM_INT4, device="cuda", seed: int = 0,
                         lm_head_qtype: Optional[int] = None) -> NativeDecoder:
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    dev = torch.device(device)
    h, i = cfg.hidden_size, cfg.intermediate_size

    def lin(n, k, qt=qtype):
        if qt in Q.T16_QTYPES:
            return Lin(_random_t16(n, k, qt, dev, gen, scale=1.0 / math.sqrt(k)))
        w = (torch.randn(n, k, device=dev, generator=gen) / math.sqrt(k)).to(torch.float16)
        return Lin(w)

    def norm():
        return (1.0 + 0.02 * torch.randn(h, device=dev, generator=gen)).to(torch.float16) - cfg.norm_offset

    layer_norm = cfg.norm == "layer"

    def vec(n):        # biases of the LayerNorm families
        return (0.02 * torch.randn(n, device=dev, generator=gen)).to(torch.float16) if layer_norm else None

    def blin(n, k, qt=qtype):
        l = lin(n, k, qt)
        l.bias = vec(n)
        return l

    embed = (torch.randn(cfg.vocab_size, h, device=dev, generator=gen) * 0.02).to(torch.float16)
    layers = []
    for _ in range(cfg.num_layers):
        if cfg.mla_kv_lora:
            # latent attention: the padding rows of q_b / kv_b are random here instead of zero (same cost, timing only)
            lat, ql, r = cfg.mla_kv_lora, cfg.mla_q_lora, cfg.rot_dim
            ones = lambda n: (1.0 + 0.02 * torch.randn(n, device=dev, generator=gen)).to(torch.float16)   # noqa: E731
            mla = MLAWeights(a=lin(ql + lat + r, h), kv_a_norm=ones(lat), kv_b=lin(2 * cfg.q_size, lat), kv_lora=lat,
                             q_lora=ql, q=None if ql else lin(cfg.q_size, h), q_a_norm=ones(ql) if ql else None,
                             q_b=lin(cfg.q_size, ql) if ql else None)
            lw = LayerWeights(input_norm=norm(), qkv=None, mla=mla, o=lin(h, cfg.q_size), post_norm=norm())
        else:
            lw = LayerWeights(input_norm=norm(), qkv=blin(cfg.q_size + 2 * cfg.kv_size, h), o=blin(h, cfg.q_size),
                              post_norm=norm(), input_norm_b=vec(h), post_norm_b=vec(h))
        if layer_norm and not cfg.gated_mlp:
            lw.up, lw.down = blin(i, h), blin(h, i)
            layers.append(lw)
            continue
        if cfg.num_experts:
            from .moe import MoEWeights
            e, mi = cfg.num_experts, cfg.moe_intermediate_size
            assert qtype in Q.T16_QTYPES, "random MoE decoders are built in a device-tiled format"
            shared = None
            shared_router = None
            if cfg.shared_expert_intermediate_size:
                si = cfg.shared_expert_intermediate_size
                shared = (lin(si, h), lin(si, h), lin(h, si))
                if cfg.model_type != "deepseek_v2":          # qwen2-moe gates its shared expert, deepseek adds it as is
                    shared_router = torch.randn(1, h, device=dev, generator=gen) / math.sqrt(h)
            lw.moe = MoEWeights(
                torch.randn(e, h, device=dev, generator=gen) / math.sqrt(h),
                _random_t16(e * mi, h, qtype, dev, gen, scale=1.0 / math.sqrt(h)).data,
                _random_t16(e * mi, h, qtype, dev, gen, scale=1.0 / math.sqrt(h)).data,
                _random_t16(e * h, mi, qtype, dev, gen, scale=1.0 / math.sqrt(mi)).data,
                qtype, e, cfg.top_k, h, mi, act=cfg.act, normalize=cfg.norm_topk_prob, shared=shared,
                shared_router=shared_router)
        else:
            lw.gate, lw.up, lw.down = lin(i, h), lin(i, h), lin(h, i)
        layers.append(lw)
    lm = lin(cfg.vocab_size, h, lm_head_qtype or qtype)
    pos_embed = None
    if cfg.no_rope and not cfg.alibi:
        pos_embed = (torch.randn(cfg.max_position + cfg.pos_offset, h, device=dev, generator=gen) * 0.02).to(torch.float16)
    return NativeDecoder(cfg, embed, layers, norm(), lm, device=device, final_norm_b=vec(h), pos_embed=pos_embed,
                         embed_norm=(norm(), vec(h)) if cfg.model_type == "bloom" else None)


MIXTRAL_8X7B = dict(vocab_size=32000, hidden_size=4096, intermediate_size=14336, num_layers=32, num_heads=32,
                    num_kv_heads=8, head_dim=128, rms_eps=1e-5, max_position=32768, rope_theta=1e6, model_type="mixtral",
                    num_experts=8, top_k=2, moe_intermediate_size=14336)
LLAMA2_7B = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_layers=32, num_heads=32,
                 num_kv_heads=32, head_dim=128, rms_eps=1e-5, max_position=4096, model_type="llama")
LLAMA3_8B = dict(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_layers=32, num_heads=32,
                 num_kv_heads=8, head_dim=128, rms_eps=1e-5, max_position=8192, rope_theta=500000.0, model_type="llama")
QWEN2_7B = dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_layers=28, num_heads=28,
                num_kv_heads=4, head_dim=128, rms_eps=1e-6, max_position=32768, rope_theta=1e6, model_type="qwen2")
TINY_LLAMA = dict(vocab_size=32000, hidden_size=2048, intermediate_size=5632, num_layers=22, num_heads=32,
                  num_kv_heads=4, head_dim=64, rms_eps=1e-5, max_position=2048, model_type="llama")
# MiniCPM3-4B: latent attention, 64 + 32 = 96 channels per head, values 64 wide (padded to 96)
MINICPM3_4B = dict(vocab_size=73448, hidden_size=2560, intermediate_size=6400, num_layers=62, num_heads=40, num_kv_heads=40,
                   head_dim=96, rot_dim=32, rope_interleaved=True, rms_eps=1e-5, max_position=32768, model_type="minicpm3",
                   mla_q_lora=768,
                   mla_kv_lora=256, attn_scale=96 ** -0.5, embed_scale=12.0, logit_scale=256 / 2560,
                   tie_word_embeddings=True)
# DeepSeek-V2-Lite: latent attention (128 + 64 -> 256-channel heads) + 64 routed experts (top 6) + 2 shared
DEEPSEEK_V2_LITE = dict(vocab_size=102400, hidden_size=2048, intermediate_size=10944, num_layers=27, num_heads=16,
                        num_kv_heads=16, head_dim=256, rot_dim=64, rope_interleaved=True, rms_eps=1e-6, max_position=32768,
                        model_type="deepseek_v2", mla_kv_lora=512, attn_scale=192 ** -0.5, num_experts=64, top_k=6,
                        moe_intermediate_size=1408, shared_expert_intermediate_size=2816, norm_topk_prob=False)
BLOOM_7B = dict(vocab_size=250880, hidden_size=4096, intermediate_size=16384, num_layers=30, num_heads=32, num_kv_heads=32,
                head_dim=128, rms_eps=1e-5, max_position=2048, model_type="bloom", norm="layer", gated_mlp=False,
                act="gelu_tanh", alibi=True, tie_word_embeddings=True)
PYTHIA_6_9B = dict(vocab_size=50432, hidden_size=4096, intermediate_size=16384, num_layers=32, num_heads=32, num_kv_heads=32,
                   head_dim=128, rms_eps=1e-5, max_position=2048, model_type="gpt_neox", norm="layer", gated_mlp=False,
                   act="gelu", parallel_residual=True, rot_dim=32)
FALCON_7B = dict(vocab_size=65024, hidden_size=4544, intermediate_size=18176, num_layers=32, num_heads=71, num_kv_heads=1,
                 head_dim=64, rms_eps=1e-5, max_position=2048, model_type="falcon", norm="layer", gated_mlp=False,
                 act="gelu", parallel_residual=True)
STARCODER_15B = dict(vocab_size=49152, hidden_size=6144, intermediate_size=24576, num_layers=40, num_heads=48, num_kv_heads=1,
                     head_dim=128, rms_eps=1e-5, max_position=8192, model_type="gpt_bigcode", norm="layer", gated_mlp=False,
                     act="gelu_tanh", no_rope=True)
