"""Conversion utilities: checkpoint discovery, imatrix reader, per-tensor mixed-quant policy, KV policy.

Capability map to the reference python/llm/src/ipex_llm/transformers/utils.py: checkpoint discovery :61-130,
imatrix reader :186-231, ``get_cur_qtype_and_imatrix`` :286-356, ``check_hidden_size`` :380-398.
"""
from __future__ import annotations

import logging
import os
import struct

import torch

from .. import qtypes as Q
from ..utils.common import invalidInputError

logger = logging.getLogger("bigdl_amd")

WEIGHTS_NAME = "pytorch_model.bin"
WEIGHTS_INDEX_NAME = "pytorch_model.bin.index.json"
SAFE_WEIGHTS_NAME = "model.safetensors"
SAFE_WEIGHTS_INDEX_NAME = "model.safetensors.index.json"


def extract_local_archive_file(pretrained_model_name_or_path, subfolder="", variant=None):
    """Locate the weight file(s) of a local checkpoint directory: returns (path, is_sharded)."""
    base = os.path.join(str(pretrained_model_name_or_path), subfolder)

    def v(name):
        if variant:
            stem, ext = name.rsplit(".", 1)
            return f"{stem}.{variant}.{ext}"
        return name
    for name, sharded in ((SAFE_WEIGHTS_NAME, False), (SAFE_WEIGHTS_INDEX_NAME, True), (WEIGHTS_NAME, False),
                          (WEIGHTS_INDEX_NAME, True)):
        f = os.path.join(base, v(name))
        if os.path.isfile(f):
            return f, sharded
    invalidInputError(False, f"Error no file named {WEIGHTS_NAME} / {SAFE_WEIGHTS_NAME} found in directory {base}")


def load_state_dict(checkpoint_file):
    if str(checkpoint_file).endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(checkpoint_file)
    return torch.load(checkpoint_file, map_location="cpu", weights_only=True)


def get_local_shard_files(pretrained_model_name_or_path, index_filename, subfolder=""):
    import json
    with open(index_filename) as f:
        index = json.load(f)
    shard_filenames = sorted(set(index["weight_map"].values()))
    sharded_metadata = index.get("metadata", {})
    sharded_metadata["all_checkpoint_keys"] = list(index["weight_map"].keys())
    sharded_metadata["weight_map"] = index["weight_map"].copy()
    shard_filenames = [os.path.join(str(pretrained_model_name_or_path), subfolder, f) for f in shard_filenames]
    return shard_filenames, sharded_metadata


def get_autocast_dtype(x_or_device):
    dev = x_or_device if isinstance(x_or_device, str) else x_or_device.device.type
    if torch.is_autocast_enabled(dev):
        return torch.get_autocast_dtype(dev)
    return None


def get_device_name(device=None) -> str:
    """Reference op xe_linear.get_xpu_device_name (transformers/utils.py:176-183): here the only target is gfx950."""
    if torch.cuda.is_available():
        return getattr(torch.cuda.get_device_properties(device or 0), "gcnArchName", "gfx950").split(":")[0]
    return "cpu"


get_xpu_device_name = get_device_name


def load_imatrix_data(imatrix_file):
    """llama.cpp imatrix file -> {tensor name: fp32 importance per input feature}.
    Layout: int32 n_entries, then per entry: int32 name_len, name bytes, int32 ncall, int32 nval, nval fp32."""
    imatrix = {}
    with open(imatrix_file, "rb") as f:
        n_entries = struct.unpack("i", f.read(4))[0]
        invalidInputError(n_entries >= 1, "failed reading name for entry from imatrix file.")
        for _ in range(n_entries):
            cur_len = struct.unpack("i", f.read(4))[0]
            cur_name = f.read(cur_len).decode("utf-8")
            ncall = struct.unpack("i", f.read(4))[0]
            nval = struct.unpack("i", f.read(4))[0]
            invalidInputError(nval >= 1, f"failed reading number of values for entry {cur_name}")
            vals = torch.frombuffer(bytearray(f.read(4 * nval)), dtype=torch.float32).clone()
            if ncall > 0:
                vals = vals / ncall
            # blk.14.attn_output.weight -> 14_o ; keyed the way get_cur_qtype_and_imatrix looks tensors up
            parts = cur_name.split(".")
            if len(parts) >= 3 and parts[0] == "blk":
                key = {"attn_q": "q", "attn_k": "k", "attn_v": "v", "attn_output": "o", "ffn_gate": "gate",
                       "ffn_up": "up", "ffn_down": "down", "attn_qkv": "qkv", "ffn_gate_inp": "gate_inp"}.get(parts[2], parts[2])
                if len(parts) >= 5 and parts[3].isdigit():          # MoE expert tensors: blk.L.ffn_down.E.weight
                    key = f"{key}_{parts[3]}"
                imatrix[f"{parts[1]}_{key}"] = vals
            else:
                imatrix[cur_name] = vals
    return imatrix


_PROJ_KEYS = {"q_proj": "q", "k_proj": "k", "v_proj": "v", "o_proj": "o", "gate_proj": "gate", "up_proj": "up",
              "down_proj": "down", "qkv_proj": "qkv", "w1": "gate", "w2": "down", "w3": "up", "dense_h_to_4h": "up",
              "dense_4h_to_h": "down", "query_key_value": "qkv", "dense": "o"}


def _layer_and_role(full_module_name: str):
    parts = full_module_name.split(".")
    layer = next((int(p) for p in parts if p.isdigit()), None)
    role = _PROJ_KEYS.get(parts[-1])
    expert = None
    if "experts" in parts:
        i = parts.index("experts")
        if i + 1 < len(parts) and parts[i + 1].isdigit():
            expert = int(parts[i + 1])
            # the first digit found was the layer; make sure we did not pick the expert index
            digits = [int(p) for p in parts if p.isdigit()]
            layer = digits[0] if digits else layer
    return layer, role, expert


def get_cur_qtype_and_imatrix(qtype, full_module_name, imatrix_data, model_config=None):
    """Per-tensor format for the mixed presets, and the importance vector of this tensor (if any).

    * gguf_iq2_xxs / gguf_iq2_xs / gguf_iq1_s: v projections in q2_k (sym_int4 without imatrix for the 1-bit
      format), `down` of the first eighth of the layers in q2_k, lm_head in sym_int8 unless an imatrix covers it
    * gguf_q4k_m: v / down of the first half of the layers in q6_k, everything else q4_k, lm_head q6_k
    * gguf_q4k_s: v / down of the first eighth of the layers in q5_k, rest q4_k
    """
    cur_qtype, cur_imatrix = qtype, None
    layer, role, expert = _layer_and_role(full_module_name)
    n_layers = getattr(model_config, "num_hidden_layers", None) if model_config is not None else None
    if imatrix_data is not None and layer is not None and role is not None:
        key = f"{layer}_{role}" if expert is None else f"{layer}_{role}_{expert}"
        cur_imatrix = imatrix_data.get(key)
        if cur_imatrix is None and role == "qkv":
            cur_imatrix = imatrix_data.get(f"{layer}_q")
    is_head = full_module_name.split(".")[-1] in ("lm_head", "output_layer", "embed_out")
    if qtype in (Q.IQ2_XXS, Q.IQ2_XS, Q.IQ1_S, Q.IQ1_M):
        if is_head:
            cur_qtype = Q.SYM_INT8 if cur_imatrix is None else qtype
        elif role == "v" or (role == "qkv" and False):
            cur_qtype = Q.Q2_K if qtype != Q.IQ1_S or cur_imatrix is not None else Q.SYM_INT4
        elif role == "down" and n_layers and layer is not None and layer < max(1, n_layers // 8):
            cur_qtype = Q.Q2_K
    elif qtype == Q.gguf_mixed_qtype["gguf_q4k_m"]:
        cur_qtype = Q.Q4_K
        if is_head or (role in ("v", "down") and n_layers and layer is not None and layer < n_layers // 2):
            cur_qtype = Q.Q6_K
    elif qtype == Q.gguf_mixed_qtype["gguf_q4k_s"]:
        cur_qtype = Q.Q4_K
        if role in ("v", "down") and n_layers and layer is not None and layer < max(1, n_layers // 8):
            cur_qtype = Q.Q5_K
    return cur_qtype, cur_imatrix


def check_hidden_size(qtype, hidden_size):
    """k-quants need rows that are multiples of 256; fall back to the closest 32-block format otherwise."""
    if hidden_size % 256 != 0:
        fallback = {Q.Q4_K: Q.ASYM_INT4, Q.Q5_K: Q.ASYM_INT5, Q.Q6_K: Q.SYM_INT8, Q.Q2_K: Q.ASYM_INT4,
                    Q.FP6_K: Q.FP6, Q.IQ2_XXS: Q.ASYM_INT4, Q.IQ2_XS: Q.ASYM_INT4, Q.IQ1_S: Q.ASYM_INT4,
                    Q.IQ1_M: Q.ASYM_INT4}
        if qtype in fallback:
            logger.info(f"hidden size {hidden_size} is not divisible by 256, required for "
                        f"{Q.qtype_name(qtype)} - using fallback quantization {Q.qtype_name(fallback[qtype])}.")
            return fallback[qtype]
    if qtype == Q.NF4 and hidden_size % 64 != 0:
        return Q.SYM_INT4
    if qtype == Q.WOQ_INT4 and hidden_size % 128 != 0:
        return Q.ASYM_INT4
    return qtype


def is_torch_bf16_gpu_available():
    return torch.cuda.is_available()


def use_quantize_kv_cache(linear=None, x=None, kv_group: int = 1) -> bool:
    """KV-cache precision policy (reference models/utils.py:77-99): env override, low-memory mode, else fp16.
    On MI355X there is no device-family table; fp8 is opt-in (or chosen by the benchmark/serving config)."""
    env = os.environ.get("IPEX_LLM_QUANTIZE_KV_CACHE", os.environ.get("BIGDL_QUANTIZE_KV_CACHE"))
    if env is not None:
        return env == "1"
    if os.environ.get("IPEX_LLM_LOW_MEM") == "1":
        return True
    return False


def engine_kv_mode():
    """KV precision of the native engine: False (fp16), True (e5m2) or "int4".  ``IPEX_LLM_QUANTIZE_KV_CACHE=int4``
    (or ``BIGDL_KV_CACHE_INT4=1``) selects the int4 cache, which the reference does not have; everything else follows
    ``use_quantize_kv_cache``."""
    env = os.environ.get("IPEX_LLM_QUANTIZE_KV_CACHE", os.environ.get("BIGDL_QUANTIZE_KV_CACHE"))
    if env == "int4" or os.environ.get("BIGDL_KV_CACHE_INT4") == "1":
        return "int4"
    return use_quantize_kv_cache()


def should_use_compresskv(x=None, prompt_len: int = 0) -> bool:
    """SnapKV policy (reference models/utils.py:317-330): env override only; never in performance mode."""
    if os.environ.get("IPEX_LLM_PERFORMANCE_MODE") == "1":
        return False
    return os.environ.get("IPEX_LLM_COMPRESS_KV_CACHE") == "1"
