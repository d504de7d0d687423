"""HF model -> low-bit model conversion.

Capability map to the reference python/llm/src/ipex_llm/transformers/convert.py:
``is_linear_module`` :138-232, GPTQ/AWQ repack :298-453, ``_replace_with_low_bit_linear`` :469-748,
``_optimize_pre`` :886-1074, ``ggml_convert_low_bit`` :1077-1169, ``convert_forward`` :1183-1205,
``_optimize_post`` :1251-2027.

Design difference: the reference binds ~45 per-architecture forwards; here the structural rewrites
(``merge_qkv``) and forward bindings are written once against the common HF decoder layout
(``bigdl_amd/models/hf_patch.py``) and the decode hot path is handed to the native engine
(``bigdl_amd/models/decoder.py``), so a model type only needs an entry in ``models/arch.py``.
"""
from __future__ import annotations

import logging
import os
import warnings

import torch
from torch import nn

from .. import qtypes as Q
from ..utils.common import invalidInputError
from .utils import check_hidden_size, get_cur_qtype_and_imatrix

logger = logging.getLogger("bigdl_amd")

_USE_VLLM = False


def get_use_vllm():
    return _USE_VLLM


def get_enable_ipex():
    """The reference's CPU IPEX/TorchScript mode (``BIGDL_OPT_IPEX``, convert.py:1208-1248) has no counterpart on
    this stack; callers that branch on it always take the regular path."""
    return False


def is_auto_gptq_available():
    try:
        import auto_gptq  # noqa: F401
        return True
    except Exception:
        return False


def is_gptq_linear(module) -> bool:
    """AutoGPTQ / optimum QuantLinear (by duck-typing: packed qweight + qzeros + scales + bits)."""
    return all(hasattr(module, a) for a in ("qweight", "qzeros", "scales", "bits", "group_size")) and \
        not _is_awq_linear(module)


def _is_awq_linear(module) -> bool:
    from .awq.linear import WQLinear_GEMM, WQLinear_GEMV
    return isinstance(module, (WQLinear_GEMM, WQLinear_GEMV)) or type(module).__name__ in ("WQLinear_GEMM",)


def is_linear_module(module):
    """-> (is_linear, (in_features, out_features, mp_group)).  Recognises nn.Linear, GPTQ/AWQ packed linears,
    DeepSpeed LinearLayer/LinearAllreduce (tensor-parallel group extracted) and vLLM parallel linears."""
    from .low_bit_linear import LowBitLinear
    in_features = out_features = mp_group = None
    is_awq = _is_awq_linear(module)
    if isinstance(module, LowBitLinear):
        return False, (module.in_len, module.out_len, module.mp_group)
    if is_gptq_linear(module) or is_awq:
        return True, (module.infeatures if hasattr(module, "infeatures") else module.in_features,
                      module.outfeatures if hasattr(module, "outfeatures") else module.out_features, None)
    if isinstance(module, nn.Linear):
        name = type(module).__name__.lower()
        if type(module).forward is not nn.Linear.forward and ("router" in name or "gate" in name):
            # MoE routers implemented as nn.Linear subclasses return (logits, weights, ids) tuples: keep their class
            return False, (module.in_features, module.out_features, None)
        return True, (module.in_features, module.out_features, None)
    cls = type(module).__name__
    if cls == "Conv1D" and hasattr(module, "nf") and getattr(module, "weight", None) is not None and module.weight.dim() == 2:
        # GPT-2 style transposed linear: weight [in, out]
        return True, (module.weight.shape[0], module.weight.shape[1], None)
    if os.environ.get("IPEX_LLM_NOT_USE_VLLM") is None and cls in (
            "ColumnParallelLinear", "RowParallelLinear", "QKVParallelLinear", "MergedColumnParallelLinear", "ParallelLMHead"):
        # vLLM tensor-parallel linears (IPEX_LLM_NOT_USE_VLLM set: never treat modules as vLLM's, reference convert.py:73): per-partition sizes, TP group only for row-parallel outputs
        if cls == "ParallelLMHead":
            return True, (module.embedding_dim, module.num_embeddings_per_partition, None)
        in_features = getattr(module, "input_size_per_partition", getattr(module, "input_size", None))
        out_features = getattr(module, "output_size_per_partition", getattr(module, "output_size", None))
        tp_size = getattr(module, "tp_size", 1)
        if cls == "RowParallelLinear" and tp_size >= 2:
            try:
                from vllm.distributed.parallel_state import get_tp_group
                mp_group = get_tp_group()
            except Exception:
                mp_group = None
        return True, (in_features, out_features, mp_group)
    if cls in ("LinearLayer", "LinearAllreduce") and hasattr(module, "weight"):
        # DeepSpeed AutoTP modules
        out_features, in_features = module.weight.shape[0], module.weight.shape[1]
        mp_group = getattr(module, "mp_group", None) if cls == "LinearAllreduce" else None
        return True, (in_features, out_features, mp_group)
    return False, (in_features, out_features, mp_group)


def is_lm_head(name, model_config, out_features):
    if name in ("lm_head", "output_layer", "embed_out"):
        return True
    vocab = getattr(model_config, "vocab_size", None) if model_config is not None else None
    # (a merged qkv / gate-up projection whose width happens to equal the vocabulary is not an output head)
    return vocab is not None and out_features == vocab and name not in ("embed_tokens", "word_embeddings") \
        and not name.endswith("_proj") and "qkv" not in name


# --------------------------------------------------------------------------------------------
# GPTQ / AWQ -> asym_int4 (ggml q4_1) re-blocking
# --------------------------------------------------------------------------------------------
_AWQ_REVERSE_ORDER = [0, 4, 1, 5, 2, 6, 3, 7]


def _unpack_int32(packed: torch.Tensor, bits: int, dim: int, order=None) -> torch.Tensor:
    """Unpack the little-endian `bits`-wide fields of a 2-D int32 tensor along `dim`.
    `order[k]` = index of the field that holds logical element k (AWQ interleaves its fields)."""
    per = 32 // bits
    shifts = torch.arange(0, 32, bits, dtype=torch.int32, device=packed.device)
    if order is not None:
        shifts = shifts[torch.tensor(order, device=packed.device)]
    p = packed.to(torch.int32)
    mask = (1 << bits) - 1
    if dim == 0:
        x = torch.bitwise_and(torch.bitwise_right_shift(p.unsqueeze(1), shifts.view(1, per, 1)), mask)
        return x.reshape(p.shape[0] * per, p.shape[1]).to(torch.uint8)
    x = torch.bitwise_and(torch.bitwise_right_shift(p.unsqueeze(2), shifts.view(1, 1, per)), mask)
    return x.reshape(p.shape[0], p.shape[1] * per).to(torch.uint8)


def convert_gptq(module, awq=False, llm_awq=False, act_order=False):
    """Packed 4-bit GPTQ/AWQ linear -> flat asym_int4 (q4_1) block buffer ``[d fp16][m=-z*d fp16][16 B codes]``.

    GPTQ : qweight [in/8, out] (fields along in), qzeros [groups, out/8] (stored zero - 1), scales [groups, out]
    AWQ  : qweight [in, out/8] and qzeros [groups, out/8] with AWQ's interleaved field order, scales [groups, out]
    Returns (buffer uint8, g_idx_map or None).
    """
    bits = module.bits if hasattr(module, "bits") else module.w_bit
    invalidInputError(bits == 4, "only 4-bit GPTQ/AWQ checkpoints can be re-blocked to asym_int4")
    group_size = module.group_size
    scales = module.scales.data
    g_idx_map = None
    if awq:
        order = None if llm_awq else _AWQ_REVERSE_ORDER
        codes = _unpack_int32(module.qweight.data, 4, 1, order)            # [in, out]
        zeros = _unpack_int32(module.qzeros.data, 4, 1, order).to(torch.float32)  # [groups, out]
        if llm_awq:                      # llm-awq stores the transposed matrix: qweight [out, in/8]
            codes = codes.t()
            zeros = zeros.t()
            scales = scales.t()
    else:
        codes = _unpack_int32(module.qweight.data, 4, 0)                    # [in, out]
        zeros = _unpack_int32(module.qzeros.data, 4, 1).to(torch.float32) + 1.0  # [groups, out]
        if act_order:
            invalidInputError(module.g_idx.shape[0] == codes.shape[0], "g_idx and weight shape mismatch")
            _, g_idx_map = torch.sort(module.g_idx)
            codes = codes[g_idx_map, :]
    in_features, out_features = codes.shape
    zeros = zeros[:, :out_features]
    invalidInputError(group_size % 32 == 0 and in_features % 32 == 0, "group size must be a multiple of 32")
    rep = group_size // 32
    codes = codes.t().contiguous().view(out_features, in_features // 32, 32)           # [out, blocks, 32]
    packed = codes[:, :, :16] | (codes[:, :, 16:] << 4)                                   # ggml nibble order
    d = scales.t().to(torch.float16).repeat_interleave(rep, dim=1)                        # [out, blocks]
    z = zeros.t().repeat_interleave(rep, dim=1)
    m = (-(z * d.float())).to(torch.float16)
    blk = torch.cat([d.contiguous().view(torch.uint8).view(out_features, -1, 2),
                     m.contiguous().view(torch.uint8).view(out_features, -1, 2), packed], dim=-1)
    return blk.reshape(-1).contiguous(), g_idx_map


def use_scale_search(model_config, qtype):
    mt = getattr(model_config, "model_type", None)
    if qtype == Q.FP6 and mt not in ("qwen2",):
        return True
    if qtype == Q.FP8E4 and mt not in ("qwen2", "baichuan"):
        name = str(getattr(model_config, "_name_or_path", "")).lower()
        if mt == "llama" and getattr(model_config, "vocab_size", 0) == 128256 and "instruct" in name:
            return False
        return True
    return False


_WEIGHT_DTYPE_READERS = {}


def _parent_reads_weight_dtype(parent, name) -> bool:
    """True when ``parent.forward`` casts something to ``self.<name>.weight.dtype`` (HF's Mamba / Mamba2 / Jamba heads do:
    ``self.lm_head(hidden_states.to(self.lm_head.weight.dtype))``): with packed low-bit blocks that dtype is the byte
    type and the cast would destroy the activations, so such a linear is left as it is.  T5-style feed-forward blocks
    read ``wo.weight.dtype`` too but get a forward without the cast instead (``_rebind_t5_feed_forward``)."""
    import inspect
    cls = type(parent)
    key = (cls, name)
    hit = _WEIGHT_DTYPE_READERS.get(key)
    if hit is None:
        try:
            hit = f"self.{name}.weight.dtype" in inspect.getsource(cls.forward)
        except (OSError, TypeError):
            hit = False
        _WEIGHT_DTYPE_READERS[key] = hit
    if hit and name == "wo" and hasattr(parent, "act") and hasattr(parent, "dropout") and (
            hasattr(parent, "wi") or hasattr(parent, "wi_0")):
        return False
    return hit


def _replace_with_low_bit_linear(model, qtype, modules_to_not_convert=None, convert_shape_only=False,
                                 cpu_embedding=False, prefix_name="", imatrix_data=None, embedding_qtype=None,
                                 model_config=None, torch_dtype=torch.float32, mixed_precision=False, act_order=False,
                                 enable_scale_search=False):
    from .embedding import CPUEmbedding, LowBitEmbedding
    from .low_bit_experts import LowBitExperts, is_fused_experts
    from .low_bit_linear import BF16Linear, FP4Params, FP16Linear, LowBitLinear
    has_been_replaced = False
    modules_to_not_convert = modules_to_not_convert or []
    for name, module in model.named_children():
        is_linear, linear_args = is_linear_module(module)
        full_module_name = prefix_name + "." + name if prefix_name else name
        if any(key in full_module_name for key in modules_to_not_convert):
            continue
        if getattr(module, "_bigdl_keep_dense", False):
            continue                          # e.g. the rank-r adapter matrices of a partial LoRA (models/remote_code.py)
        if is_linear and _parent_reads_weight_dtype(model, name):
            logger.info("%s stays dense: %s casts activations to its weight's dtype", full_module_name, type(model).__name__)
            continue
        if is_linear and not isinstance(module, LowBitLinear):
            in_features, out_features, mp_group = linear_args
            # (not for encoder-decoder models: a teacher-forced pass scores every decoder position, and a decoding
            # step has one position anyway)
            optimize_lm_head = is_lm_head(name, model_config, out_features) and \
                os.environ.get("IPEX_LLM_LAST_LM_HEAD", None) != "0" and \
                not getattr(model_config, "is_encoder_decoder", False)
            new_linear = None
            packed = is_gptq_linear(module) or _is_awq_linear(module)
            if packed:
                is_awq = _is_awq_linear(module)
                has_bias = module.bias is not None and module.bias.abs().sum() != 0
                device = module.qweight.data.device
                invalidInputError(device.type != "meta", "converting from meta device is not supported")
                new_linear = LowBitLinear(in_features, out_features, qtype=qtype, bias=has_bias, mp_group=mp_group,
                                          optimize_lm_head=optimize_lm_head, act_order=act_order,
                                          enable_scale_search=enable_scale_search)
                weight, g_idx_map = convert_gptq(module, awq=is_awq, act_order=act_order,
                                                 llm_awq=is_awq and getattr(module, "backend", "") == "llm-awq")
                if act_order:
                    new_linear.g_idx_map = g_idx_map
                new_linear._parameters["weight"] = FP4Params(data=weight.cpu(), requires_grad=False, quantized=True,
                                                             _shape=(out_features, in_features), qtype=qtype).to(device)
                if has_bias:
                    new_linear._parameters["bias"] = nn.Parameter(module.bias.data).to(device)
            elif qtype not in (Q.FP16, Q.BF16):
                if in_features % 64 != 0:
                    continue   # kernels need whole 32-blocks per MFMA step; such layers stay dense
                cur_qtype, cur_imatrix = get_cur_qtype_and_imatrix(qtype, full_module_name, imatrix_data, model_config)
                if mixed_precision and is_lm_head(name, model_config, out_features) and \
                        cur_qtype in (Q.SYM_INT4, Q.ASYM_INT4):
                    cur_qtype = Q.Q6_K
                cur_qtype = check_hidden_size(cur_qtype, in_features)
                device = module.weight.data.device
                new_linear = LowBitLinear(in_features, out_features, cur_qtype, module.bias is not None,
                                          mp_group=mp_group, optimize_lm_head=optimize_lm_head,
                                          enable_scale_search=enable_scale_search)
                src_weight = module.weight.data
                if type(module).__name__ == "Conv1D":
                    src_weight = src_weight.t().contiguous() if src_weight.device.type != "meta" else \
                        torch.empty(out_features, in_features, dtype=src_weight.dtype, device="meta")
                new_linear._parameters["weight"] = FP4Params(
                    data=src_weight, requires_grad=False, quantized=False, _shape=None,
                    convert_shape_only=convert_shape_only, qtype=cur_qtype, imatrix=cur_imatrix,
                    in_features=in_features, enable_scale_search=enable_scale_search).to(device)
                if module.bias is not None:
                    new_linear._parameters["bias"] = nn.Parameter(module.bias.data).to(device)
            else:
                cls, dt = (FP16Linear, torch.float16) if qtype == Q.FP16 else (BF16Linear, torch.bfloat16)
                module.to(dt)
                new_linear = cls(in_features, out_features, module.bias is not None, mp_group=mp_group,
                                 optimize_lm_head=optimize_lm_head)
                dense_w = module.weight.data.t().contiguous() if type(module).__name__ == "Conv1D" else module.weight.data
                new_linear._parameters["weight"] = nn.Parameter(dense_w, requires_grad=False)
                if module.bias is not None:
                    new_linear._parameters["bias"] = nn.Parameter(module.bias.data, requires_grad=False)
            if new_linear is not None:
                if not module.training:
                    new_linear.eval()
                model._modules[name] = new_linear
                has_been_replaced = True
                model._modules[name].requires_grad_(False)
                module.weight = None
        elif is_fused_experts(module) and qtype not in (Q.FP16, Q.BF16) and module.gate_up_proj.shape[-1] % 64 == 0 \
                and module.down_proj.shape[-1] % 64 == 0:
            cur_qtype, cur_imatrix = get_cur_qtype_and_imatrix(qtype, full_module_name, None, model_config)
            # one format for the three stacked projections: both reduction sizes (hidden for gate / up, intermediate
            # for down) have to fit its block size
            cur_qtype = check_hidden_size(check_hidden_size(cur_qtype, module.gate_up_proj.shape[-1]), module.down_proj.shape[-1])
            if check_hidden_size(cur_qtype, module.gate_up_proj.shape[-1]) != cur_qtype:
                cur_qtype = Q.ASYM_INT4
            # (the routed kernels read all experts of a layer in one format: a mixture resolves to its integer member)
            cur_qtype = {Q.MOFQ4: Q.SYM_INT4, Q.MOFQ8: Q.SYM_INT8}.get(cur_qtype, cur_qtype)
            model._modules[name] = LowBitExperts.from_experts(module, cur_qtype, convert_shape_only,
                                                              enable_scale_search=enable_scale_search)
            has_been_replaced = True
            continue
        elif cpu_embedding and type(module) == nn.Embedding:
            model._modules[name] = CPUEmbedding.from_embedding(module)
        elif embedding_qtype is not None and type(module) == nn.Embedding:
            model._modules[name] = LowBitEmbedding.from_embedding(module, convert_shape_only, embedding_qtype)
        if len(list(module.children())) > 0:
            _, flag = _replace_with_low_bit_linear(
                module, qtype, modules_to_not_convert, convert_shape_only, cpu_embedding,
                prefix_name=full_module_name, imatrix_data=imatrix_data, embedding_qtype=embedding_qtype,
                model_config=model_config, torch_dtype=torch_dtype, mixed_precision=mixed_precision,
                act_order=act_order, enable_scale_search=enable_scale_search)
            has_been_replaced = flag or has_been_replaced
    return model, has_been_replaced


def convert_forward(m, target_m, new_forward):
    """Rebind ``forward`` of every sub-module whose class is ``target_m``."""
    import types
    if m.__class__ == target_m:
        m.forward = types.MethodType(new_forward, m)
    for _, sub_m in m.named_children():
        convert_forward(sub_m, target_m, new_forward)


def replace_func(m, target_m, func_name, new_func):
    import types
    for _, sub_m in m.named_children():
        if sub_m.__class__ == target_m:
            setattr(sub_m, func_name, types.MethodType(new_func, sub_m))
        replace_func(sub_m, target_m, func_name, new_func)


def convert_bigdl_other_module(model, dtype):
    """Everything that is not a quantized linear (norms, embeddings, biases, buffers) goes to ``dtype``."""
    from .low_bit_linear import BF16Linear, FP16Linear, LowBitLinear
    for module in model.modules():
        if list(module.children()):
            continue
        if isinstance(module, (LowBitLinear, FP16Linear, BF16Linear)):
            if isinstance(module, LowBitLinear) and module.bias is not None:
                module.bias.data = module.bias.data.to(dtype)
            continue
        module.to(dtype)


def _optimize_pre(model, qtype=None):
    from ..models import hf_patch
    return hf_patch.optimize_pre(model, qtype)


def _optimize_post(model):
    from ..models import hf_patch
    return hf_patch.optimize_post(model)


def ggml_convert_low_bit(model, qtype, optimize_model=True, convert_shape_only=False, device="cpu",
                         modules_to_not_convert=None, cpu_embedding=False, torch_dtype="auto", imatrix_data=None,
                         embedding_qtype=None, mixed_precision=False, disable_optimize_pre=False):
    """Top-level conversion: structural rewrites -> quantize every linear -> cast the rest -> bind forwards."""
    logger.info(f"Converting the current model to {Q.qtype_name(qtype) if qtype in Q._id_to_name else qtype} format......")
    modules_to_not_convert = [] if modules_to_not_convert is None else list(modules_to_not_convert)
    model_config = getattr(model, "config", None)
    enable_scale_search = use_scale_search(model_config, qtype) if model_config is not None else False
    act_order = False
    if getattr(model, "quantization_method", None) == "gptq":
        act_order = bool(getattr(getattr(model_config, "quantization_config", None), "desc_act", False)) or \
            bool((getattr(model_config, "quantization_config", None) or {}).get("desc_act", False)
                 if isinstance(getattr(model_config, "quantization_config", None), dict) else False)
    if optimize_model and not disable_optimize_pre:
        model = _optimize_pre(model, qtype)
    model, has_been_replaced = _replace_with_low_bit_linear(
        model, qtype, modules_to_not_convert, convert_shape_only, cpu_embedding, imatrix_data=imatrix_data,
        embedding_qtype=embedding_qtype, model_config=model_config, torch_dtype=torch_dtype,
        mixed_precision=mixed_precision, act_order=act_order, enable_scale_search=enable_scale_search)
    if not has_been_replaced:
        warnings.warn("No linear modules were found in your model, it will be keeped as it is.")
    elif device == "cpu":
        if torch_dtype not in ("auto", None) and not getattr(model, "quantization_method", None) == "gptq":
            convert_bigdl_other_module(model, torch_dtype if isinstance(torch_dtype, torch.dtype) else torch.float32)
        elif torch_dtype == "auto" or torch_dtype is None:
            convert_bigdl_other_module(model, torch.float32)
    elif device == "meta":
        pass
    _rebind_t5_feed_forward(model)
    if optimize_model:
        model = _optimize_post(model)
    return model


def _t5_gated_ff_forward(self, hidden_states):
    return self.wo(self.dropout(self.act(self.wi_0(hidden_states)) * self.wi_1(hidden_states)))


def _t5_ff_forward(self, hidden_states):
    return self.wo(self.dropout(self.act(self.wi(hidden_states))))


def _rebind_t5_feed_forward(model):
    """T5-style feed-forward blocks (t5, mt5, umt5, longt5, switch_transformers, pix2struct text) cast their hidden
    states to ``wo.weight.dtype`` before the output projection, which for a low-bit ``wo`` is the byte type of the
    packed blocks.  The reference works around it by leaving ``wo`` unconverted
    (example/GPU/HuggingFace/LLM/flan-t5/generate.py:44-52); here such a block gets a forward without the cast."""
    import types
    from .low_bit_linear import LowBitLinear
    for m in model.modules():
        if not isinstance(getattr(m, "wo", None), LowBitLinear) or not hasattr(m, "act") or not hasattr(m, "dropout"):
            continue
        if hasattr(m, "wi_0") and hasattr(m, "wi_1"):
            m.forward = types.MethodType(_t5_gated_ff_forward, m)
        elif hasattr(m, "wi"):
            m.forward = types.MethodType(_t5_ff_forward, m)
