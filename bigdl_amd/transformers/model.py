"""``AutoModel*`` front end: ``from_pretrained(load_in_4bit / load_in_low_bit ...)``, ``save_low_bit``,
``load_low_bit``, ``from_gguf``.

Capability map to the reference python/llm/src/ipex_llm/transformers/model.py: ``save_low_bit`` :59-103,
``from_pretrained`` :111-388 (kwargs, BitsAndBytes / GPTQ / AWQ detection, speculative, pipeline stages,
lookup), ``load_convert`` :410-528, ``load_low_bit`` :530-785, Auto classes :788-830.

Checkpoint format (kept): ``pytorch_model.bin`` (or ``pytorch_model-0000i-of-0000n.bin`` + index) in which every
``LowBitLinear.weight`` is the flat host-layout block buffer (uint8; bf16-typed for nf4), ``config.json`` with
``bigdl_transformers_low_bit: <qtype name>``, ``load_keys.json`` = {"all_checkpoint_keys": [...]} and
``generation_config.json``.  Loading builds the skeleton on the meta device, swaps in shape-only low-bit
linears and fills them from the files with ``torch.load(weights_only=True)``.
"""
from __future__ import annotations

import copy
import json
import logging
import os
import types
import warnings

import torch
import transformers
from transformers.configuration_utils import PretrainedConfig

from .. import qtypes as Q
from ..qtypes import ggml_tensor_qtype, gguf_mixed_qtype
from ..utils.common import invalidInputError
from .utils import extract_local_archive_file, get_local_shard_files, load_imatrix_data, load_state_dict

logger = logging.getLogger("bigdl_amd")

patched_training_mode = None
MAX_SHARD_BYTES = 5 * 1024 ** 3


def _state_dict_for_save(model):
    sd = {}
    for k, v in model.state_dict().items():
        t = v.data if isinstance(v, torch.nn.Parameter) else v
        sd[k] = t.detach()
    # an embedding that lives in a file (DiskEmbedding keeps an empty placeholder as its parameter) is written out whole
    from .embedding import DiskEmbedding
    for name, mod in model.named_modules():
        if isinstance(mod, DiskEmbedding) and f"{name}.weight" in sd:
            sd[f"{name}.weight"] = mod.to_embedding().weight.data
    # tied weights are stored once (HF convention): drop lm_head when it aliases the embedding
    tied = getattr(model.config, "tie_word_embeddings", False)
    if tied and "lm_head.weight" in sd:
        for k in list(sd):
            if k != "lm_head.weight" and sd[k].data_ptr() == sd["lm_head.weight"].data_ptr():
                del sd["lm_head.weight"]
                break
    return sd


def _write_shards(sd, save_dir, max_shard_bytes=MAX_SHARD_BYTES):
    shards, cur, cur_bytes = [], {}, 0
    for k, v in sd.items():
        nb = v.numel() * v.element_size()
        if cur and cur_bytes + nb > max_shard_bytes:
            shards.append(cur)
            cur, cur_bytes = {}, 0
        cur[k] = v
        cur_bytes += nb
    shards.append(cur)
    if len(shards) == 1:
        torch.save(shards[0], os.path.join(save_dir, "pytorch_model.bin"))
        return
    weight_map, total = {}, 0
    for i, sh in enumerate(shards):
        name = f"pytorch_model-{i + 1:05d}-of-{len(shards):05d}.bin"
        torch.save(sh, os.path.join(save_dir, name))
        for k, v in sh.items():
            weight_map[k] = name
            total += v.numel() * v.element_size()
    with open(os.path.join(save_dir, "pytorch_model.bin.index.json"), "w") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2)


def save_low_bit(self, *args, **kwargs):
    invalidInputError(self.config.to_dict().get("bigdl_transformers_low_bit", False),
                      "Detected this model is not a low-bit model, please use from_pretrained's load_in_4bit or "
                      "load_in_low_bit parameter to load a 4-bit model first.")
    save_dir = args[0] if args else kwargs["save_directory"]
    os.makedirs(save_dir, exist_ok=True)
    if hasattr(self.config, "quantization_config"):
        delattr(self.config, "quantization_config")
        for attr in ("quantization_method", "is_quantized"):
            if hasattr(self, attr):
                try:
                    delattr(self, attr)
                except AttributeError:
                    pass
    origin_device = next(self.parameters()).device
    self.to("cpu")                       # device tiles -> host block order
    architectures = getattr(self.config, "architectures", None)
    sd = _state_dict_for_save(self)
    _write_shards(sd, save_dir, kwargs.get("max_shard_size_bytes", MAX_SHARD_BYTES))
    if architectures:
        self.config.update({"architectures": architectures})
    self.config.save_pretrained(save_dir)
    gc = getattr(self, "generation_config", None)
    if gc is not None:
        try:
            gc.save_pretrained(save_dir)
        except Exception as e:  # invalid sampling combos in third-party configs must not block a save
            logger.warning(f"generation_config not saved: {e}")
    from .low_bit_linear import tensor_qtype_overrides
    with open(os.path.join(save_dir, "load_keys.json"), "w") as f:
        json.dump({"all_checkpoint_keys": list(sd.keys()), "tensor_qtypes": tensor_qtype_overrides(self)}, f)
    if origin_device.type != "cpu":
        self.to(origin_device)


def _assign_state_dict(model, state_dict, strict_missing=None):
    """Fill (possibly meta) parameters/buffers from `state_dict` by assignment; keeps FP4Params metadata."""
    from .low_bit_linear import FP4Params
    missing = []
    params = dict(model.named_parameters(remove_duplicate=False))
    buffers = dict(model.named_buffers(remove_duplicate=False))
    for name, value in state_dict.items():
        mod_name, _, leaf = name.rpartition(".")
        try:
            module = model.get_submodule(mod_name) if mod_name else model
        except AttributeError:
            continue
        if name in params:
            old = params[name]
            if isinstance(old, FP4Params):
                new = FP4Params(value, requires_grad=False, quantized=True, _shape=old._shape, qtype=old.qtype,
                                host_only=old.host_only)
            else:
                if old.device.type != "meta" and old.dtype != value.dtype and value.is_floating_point():
                    value = value.to(old.dtype)
                elif old.device.type == "meta" and value.is_floating_point() and old.is_floating_point():
                    value = value.to(old.dtype)
                new = torch.nn.Parameter(value, requires_grad=False)
            module._parameters[leaf] = new
        elif name in buffers:
            module._buffers[leaf] = value
    for name, p in model.named_parameters(remove_duplicate=False):
        if p.device.type == "meta" and name not in state_dict:
            missing.append(name)
    return missing


def _materialize_meta_buffers(model):
    """Non-persistent buffers (rotary inv_freq ...) are not in checkpoints: recompute them by re-running the
    owning module's constructor logic where HF offers it, else zero-fill."""
    for module in model.modules():
        for bname, buf in list(module._buffers.items()):
            if buf is not None and buf.device.type == "meta":
                if bname in ("inv_freq", "original_inv_freq") and hasattr(module, "rope_init_fn"):
                    try:
                        inv, scaling = module.rope_init_fn(module.config, torch.device("cpu"))
                        module._buffers[bname] = inv
                        module.attention_scaling = scaling
                        continue
                    except Exception:
                        pass
                if bname in ("inv_freq", "original_inv_freq") and hasattr(module, "compute_default_rope_parameters"):
                    try:
                        inv, scaling = module.compute_default_rope_parameters(module.config, torch.device("cpu"))
                        module._buffers[bname] = inv
                        module.attention_scaling = scaling
                        continue
                    except Exception:
                        pass
                module._buffers[bname] = torch.zeros(buf.shape, dtype=buf.dtype, device="cpu")


def _one_rank_pipeline(stages) -> bool:
    """``pipeline_parallel_stages=1`` passed explicitly while a one-rank process group is up"""
    return stages == 1 and torch.distributed.is_available() and torch.distributed.is_initialized() \
        and torch.distributed.get_world_size() == 1


def _register_builtin_families():
    from ..models.yuan import register
    register()


_register_builtin_families()

class _BaseAutoModelClass:
    HF_MODEL = None

    @classmethod
    def from_pretrained(cls, *args, **kwargs):
        """HF ``from_pretrained`` plus ``load_in_4bit`` / ``load_in_low_bit`` / ``optimize_model`` /
        ``modules_to_not_convert`` / ``speculative`` / ``cpu_embedding`` / ``disk_embedding`` / ``imatrix`` /
        ``embedding_qtype`` / ``mixed_precision`` / ``pipeline_parallel_stages`` / ``model_hub``."""
        pretrained_model_name_or_path = kwargs.get("pretrained_model_name_or_path", None) if len(args) == 0 else args[0]
        model_hub = kwargs.pop("model_hub", "huggingface")
        invalidInputError(model_hub in ["huggingface", "modelscope"],
                          "The parameter `model_hub` is supposed to be `huggingface` or `modelscope`, "
                          f"but got {model_hub}.")
        dm = kwargs.get("device_map")
        invalidInputError(not (dm and isinstance(dm, str) and ("xpu" in dm or "cuda" in dm)),
                          "Please do not use `device_map` with a device value as an argument. "
                          "Use model.to('cuda') instead.")
        if model_hub == "modelscope":
            try:
                from modelscope.utils.hf_util import get_wrapped_class
                cls.HF_Model = get_wrapped_class(cls.HF_Model)
            except ImportError:
                invalidInputError(False, "model_hub='modelscope' needs the `modelscope` package")
        config_dict, _ = PretrainedConfig.get_config_dict(pretrained_model_name_or_path)
        bigdl_transformers_low_bit = config_dict.pop("bigdl_transformers_low_bit", False)
        native_engine = kwargs.pop("native_engine", None)
        if not bigdl_transformers_low_bit and isinstance(pretrained_model_name_or_path, str) \
                and os.path.isdir(pretrained_model_name_or_path):
            from ..models.arch import NativeCausalLM, is_native_arch
            if is_native_arch(config_dict, native_engine):
                # families HuggingFace only has as remote code (chatglm, baichuan, qwen v1, internlm, minicpm, ...):
                # loaded through their weight-name map into the native engine, no remote code is executed
                low_bit = kwargs.get("load_in_low_bit") or "sym_int4"
                return NativeCausalLM.from_pretrained(pretrained_model_name_or_path, low_bit=low_bit)
        invalidInputError(not bigdl_transformers_low_bit,
                          f"Detected model is a low-bit({bigdl_transformers_low_bit}) model, "
                          f"Please use load_low_bit to load this model.")

        load_in_4bit = kwargs.pop("load_in_4bit", False)
        load_in_low_bit = kwargs.pop("load_in_low_bit", None)
        optimize_model = kwargs.pop("optimize_model", True)
        user_quantization_config = kwargs.pop("quantization_config", None)
        speculative = kwargs.pop("speculative", False)
        pipeline_parallel_stages = kwargs.pop("pipeline_parallel_stages", 0) or 0
        torch_dtype = kwargs.pop("torch_dtype", kwargs.pop("dtype", None))
        embedding_qtype = kwargs.pop("embedding_qtype", None)

        if user_quantization_config is not None and "BitsAndBytesConfig" in str(user_quantization_config.__class__):
            bnb = user_quantization_config
            t = getattr(bnb, "bnb_4bit_quant_type", None)
            if t is not None:
                if t == "nf4":
                    load_in_low_bit = "nf4"
                elif t == "fp4":
                    warnings.warn("QLoRA does not support fp4 now, use default nf4", FutureWarning)
                    load_in_low_bit = "nf4"
                elif t == "int4":
                    load_in_low_bit = "sym_int4"
                elif t == "bf16":
                    load_in_low_bit = "bf16"
                else:
                    invalidInputError(False, "Only nf4 or int4 is supported for bnb_4bit_quant_type")
            else:
                warnings.warn("bnb_4bit_quant_type is None, use default int4", FutureWarning)
                load_in_low_bit = "sym_int4"
            if getattr(bnb, "bnb_4bit_use_double_quant", False) is True:
                warnings.warn("QLoRA does not support double quant now, set to False", FutureWarning)
            cd = getattr(bnb, "bnb_4bit_compute_dtype", None)
            if cd is not None:
                invalidInputError(cd in (torch.float32, torch.bfloat16),
                                  "Only float32 or bfloat16 is supported for bnb_4bit_compute_dtype")
                torch_dtype = cd
            else:
                warnings.warn("torch_dtype is None, use default float32", FutureWarning)
                torch_dtype = torch.float32
            optimize_model = False
            kwargs["modules_to_not_convert"] = ["lm_head"]
            user_quantization_config = None

        from ..llm_patching import bigdl_patched
        if bigdl_patched == "Train":
            global patched_training_mode
            if load_in_low_bit in ("nf4", "sym_int4") or load_in_4bit:
                patched_training_mode = "qlora"
            else:
                patched_training_mode = "lora"
                load_in_low_bit = "bf16"
            optimize_model = False
            kwargs["modules_to_not_convert"] = ["lm_head"]

        if not (load_in_4bit or load_in_low_bit):
            if torch_dtype is not None:
                kwargs["dtype"] = torch_dtype
            return cls.HF_Model.from_pretrained(*args, **kwargs)

        q_config = config_dict.get("quantization_config")
        if q_config is not None:
            from .low_bit_linear import get_block_size
            method = q_config.get("quant_method")
            if method in ("gptq", "awq"):
                invalidInputError(q_config.get("bits", 4) == 4, f"Only 4-bit {method} is supported.")
                if load_in_low_bit is not None:
                    invalidInputError(load_in_low_bit == "asym_int4",
                                      f"You can only load {method} model as aysm_int4 low bit type.")
                load_in_low_bit = "asym_int4"
                invalidInputError(int(q_config["group_size"]) % get_block_size("asym_int4") == 0,
                                  f"group_size must be divisible by {get_block_size('asym_int4')}.")
                kwargs["_packed_quant"] = (method, q_config)

        if load_in_low_bit == "fp16":
            invalidInputError(torch_dtype in (None, torch.float16),
                              "Please use torch_dtype=torch.float16 when setting load_in_low_bit='fp16'.")
            torch_dtype = torch.float16
        elif load_in_low_bit == "bf16":
            invalidInputError(torch_dtype in (None, torch.bfloat16),
                              "Please use torch_dtype=torch.bfloat16 when setting load_in_low_bit='bf16'.")
            torch_dtype = torch.bfloat16
        else:
            torch_dtype = torch_dtype or "auto"
        kwargs["torch_dtype"] = torch_dtype
        q_k = load_in_low_bit if load_in_low_bit else "sym_int4"
        invalidInputError(q_k not in ["sym_int4_rtn", "sym_int8_rtn", "asym_int4_rtn"],
                          f"The dtype {q_k} is a per-channel RTN format and cannot be used for LowBitLinear models")
        imatrix_file = kwargs.pop("imatrix", None)
        if q_k in ["gguf_iq2_xxs", "gguf_iq2_xs", "gguf_iq1_s"]:
            invalidInputError(imatrix_file is not None, "For gguf_iq2 and gguf_iq1 quantization, imatrix is needed.")
        cpu_embedding = kwargs.get("cpu_embedding", False)
        if not cpu_embedding and embedding_qtype is None:
            if q_k in ["gguf_iq2_xxs", "gguf_iq2_xs", "gguf_iq1_s", "q2_k"]:
                embedding_qtype = "q2_k"
            elif q_k in ["gguf_q4k_s", "gguf_q4k_m"]:
                embedding_qtype = "q4_k"
        if imatrix_file is not None:
            kwargs["imatrix_data"] = load_imatrix_data(imatrix_file)
        kwargs["embedding_qtype"] = embedding_qtype
        model = cls.load_convert(q_k, optimize_model, *args, **kwargs)

        # one stage under an initialized one-rank process group is a valid (degenerate) pipeline: same code path,
        # same timing attributes, no peers to talk to
        if pipeline_parallel_stages > 1 or _one_rank_pipeline(pipeline_parallel_stages):
            invalidInputError(not speculative, "Please do not set speculative=True when using pipeline_parallel_stages")
            invalidInputError(torch.distributed.is_initialized() and
                              torch.distributed.get_world_size() == pipeline_parallel_stages,
                              "Please make sure you've called `init_pipeline_parallel()` "
                              "and world size is the same as `pipeline_parallel_stages`")
            from .pipeline_parallel import pipeline_parallel, pipeline_parallel_generate
            model = pipeline_parallel(model, pipeline_parallel_stages, torch_dtype)
            model.pipeline_parallel_generate = types.MethodType(pipeline_parallel_generate, model)
            torch.distributed.barrier()
        if speculative:
            from .speculative import attach_speculative
            kw2 = dict(kwargs)
            draft_model = cls.load_convert("sym_int4", optimize_model, *args, **kw2)
            attach_speculative(model, draft_model)
        from .lookup import lookup_generate
        model.lookup_generate = types.MethodType(lookup_generate, model)
        return model

    @staticmethod
    def from_gguf(fpath: str, optimize_model: bool = True, cpu_embedding: bool = False, low_bit: str = "sym_int4"):
        """Load a gguf file -> (low-bit model, HF tokenizer)."""
        from ..optimize import optimize_model as optimize_model_fn
        from .gguf.api import load_gguf_model
        model, tokenizer = load_gguf_model(fpath, dtype=torch.half, low_bit=low_bit)
        model = optimize_model_fn(model, low_bit=low_bit, optimize_llm=optimize_model, cpu_embedding=cpu_embedding)
        return model, tokenizer

    @classmethod
    def load_convert(cls, q_k, optimize_model, *args, **kwargs):
        from .convert import ggml_convert_low_bit
        invalidInputError(q_k in ggml_tensor_qtype or q_k in gguf_mixed_qtype,
                          f"Unknown load_in_low_bit value: {q_k}, expected:"
                          f" sym_int4, asym_int4, sym_int5, asym_int5, sym_int8, nf3, nf4, fp4, fp6, fp8, fp8_e4m3,"
                          f" fp8_e5m2, fp16, bf16, gguf_iq2_xxs, gguf_iq2_xs, gguf_iq1_s, q2_k, q4_k, q5_k, q6_k, fp6_k,"
                          f" gguf_q4k_s, gguf_q4k_m, mixed_fp4 or mixed_fp8.")
        qtype = Q.qtype_id(q_k)
        modules_to_not_convert = kwargs.pop("modules_to_not_convert", None)
        mixed_precision = kwargs.pop("mixed_precision", False)
        cpu_embedding = kwargs.pop("cpu_embedding", False)
        disk_embedding = kwargs.pop("disk_embedding", False)
        kwargs.pop("use_cache", None) if False else None
        imatrix_data = kwargs.pop("imatrix_data", None)
        embedding_qtype = kwargs.pop("embedding_qtype", None)
        disable_optimize_pre = kwargs.pop("disable_optimize_pre", False)
        packed = kwargs.pop("_packed_quant", None)
        torch_dtype = kwargs.pop("torch_dtype", "auto")
        if embedding_qtype is not None:
            embedding_qtype = ggml_tensor_qtype[embedding_qtype]
        _args = copy.deepcopy(args)
        _kwargs = copy.deepcopy({k: v for k, v in kwargs.items() if k != "config"})
        if "config" in kwargs:
            _kwargs["config"] = kwargs["config"]
        _kwargs.pop("trust_remote_code", None) if False else None
        _kwargs["dtype"] = torch_dtype
        _kwargs.setdefault("low_cpu_mem_usage", True)
        awq_config = None
        if packed is not None:
            model = cls._load_packed(packed, *_args, **_kwargs)
        else:
            try:
                model = cls.HF_Model.from_pretrained(*_args, **_kwargs)
            except NotImplementedError:
                logger.info("Failed to load models with `low_cpu_mem_usage` specified, "
                            "will fall to traditional load method with higher memory consumption.")
                _kwargs["low_cpu_mem_usage"] = False
                model = cls.HF_Model.from_pretrained(*_args, **_kwargs)
                model.config.update({"bigdl_lcmu_enabled": False})
        model = model.to("cpu")
        dt = torch_dtype
        model = ggml_convert_low_bit(model, qtype, optimize_model, modules_to_not_convert=modules_to_not_convert,
                                     cpu_embedding=cpu_embedding, torch_dtype=dt, imatrix_data=imatrix_data,
                                     embedding_qtype=embedding_qtype, mixed_precision=mixed_precision,
                                     disable_optimize_pre=disable_optimize_pre)
        if disk_embedding:
            from .embedding import DiskEmbedding
            DiskEmbedding.replace_normal_embedding(model)
        model.config.update({"bigdl_transformers_low_bit": q_k, "bigdl_disk_embedding": disk_embedding,
                             # what load_low_bit has to repeat to rebuild the same module tree (its keyword arguments
                             # of the same names override these)
                             "bigdl_convert_options": {"optimize_model": bool(optimize_model),
                                                       "modules_to_not_convert": list(modules_to_not_convert or []),
                                                       "mixed_precision": bool(mixed_precision)}})
        if getattr(model.config, "tie_word_embeddings", False) and not cpu_embedding and embedding_qtype is None:
            # a head that stayed a plain nn.Linear shares its weight with the embedding again (the dtype pass above
            # made two copies); a converted head owns packed low-bit blocks, which re-tying would overwrite with the
            # dense embedding matrix
            try:
                head = model.get_output_embeddings()
            except Exception:
                head = None
            if type(head) is torch.nn.Linear:
                try:
                    model.tie_weights()
                except Exception:
                    pass
        model.save_low_bit = types.MethodType(save_low_bit, model)
        return model

    @classmethod
    def _load_packed(cls, packed, *args, **kwargs):
        """GPTQ / AWQ checkpoints: build the skeleton, swap linears for packed containers, fill from the files."""
        method, q_config = packed
        path = args[0] if args else kwargs["pretrained_model_name_or_path"]
        config = transformers.AutoConfig.from_pretrained(path, trust_remote_code=kwargs.get("trust_remote_code", False))
        if hasattr(config, "quantization_config"):
            delattr(config, "quantization_config")
        dt = kwargs.get("dtype")
        dt = torch.float16 if dt in ("auto", None) else dt
        model = cls.HF_Model.from_config(config, dtype=dt, trust_remote_code=kwargs.get("trust_remote_code", False))
        if method == "awq":
            from .awq import AwqConfig, _replace_with_awq_layers
            model = _replace_with_awq_layers(model, AwqConfig.from_dict(q_config), q_config.get("modules_to_not_convert"))
        else:
            from .gptq import replace_with_gptq_layers
            model = replace_with_gptq_layers(model, q_config)
            model.quantization_method = "gptq"
        f, sharded = extract_local_archive_file(path)
        files = get_local_shard_files(path, f)[0] if sharded else [f]
        for ck in files:
            sd = load_state_dict(ck)
            model.load_state_dict(sd, strict=False)
        model.config.quantization_config = q_config
        return model

    @classmethod
    def load_low_bit(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        """Load a directory written by ``save_low_bit``."""
        from .convert import ggml_convert_low_bit
        optimize_model = kwargs.pop("optimize_model", None)
        modules_to_not_convert = kwargs.pop("modules_to_not_convert", None)
        cpu_embedding = kwargs.pop("cpu_embedding", False)
        embedding_qtype = kwargs.pop("embedding_qtype", None)
        disk_embedding = kwargs.pop("disk_embedding", False)
        pipeline_parallel_stages = kwargs.pop("pipeline_parallel_stages", 0) or 0
        speculative = kwargs.pop("speculative", False)
        torch_dtype = kwargs.pop("torch_dtype", kwargs.pop("dtype", "auto"))
        trust_remote_code = kwargs.pop("trust_remote_code", False)
        subfolder = kwargs.pop("subfolder", "")
        variant = kwargs.pop("variant", None)
        mixed_precision = kwargs.pop("mixed_precision", None)
        config = kwargs.pop("config", None)
        if config is None and isinstance(pretrained_model_name_or_path, str) and os.path.isfile(
                os.path.join(pretrained_model_name_or_path, "config.json")):
            import json
            with open(os.path.join(pretrained_model_name_or_path, "config.json")) as f:
                if "bigdl_native_arch" in json.load(f):
                    from ..models.arch import NativeCausalLM
                    return NativeCausalLM.load_low_bit(pretrained_model_name_or_path)
        if config is None:
            config = transformers.AutoConfig.from_pretrained(pretrained_model_name_or_path,
                                                            trust_remote_code=trust_remote_code)
        config_dict = config.to_dict()
        bigdl_transformers_low_bit = config_dict.pop("bigdl_transformers_low_bit", False)
        saved = config_dict.pop("bigdl_convert_options", None) or {}
        optimize_model = saved.get("optimize_model", True) if optimize_model is None else optimize_model
        mixed_precision = saved.get("mixed_precision", False) if mixed_precision is None else mixed_precision
        if modules_to_not_convert is None:
            modules_to_not_convert = saved.get("modules_to_not_convert") or None
        invalidInputError(bigdl_transformers_low_bit,
                          "Detect this model is not a low-bit model, Please use from_pretrained"
                          " with load_in_4bit or load_in_low_bit to get a low-bit model , and "
                          " serialize the model using save_low_bit first.")
        invalidInputError(bigdl_transformers_low_bit in ggml_tensor_qtype or bigdl_transformers_low_bit in gguf_mixed_qtype,
                          f"Unknown bigdl_transformers_low_bit value: {bigdl_transformers_low_bit}")
        qtype = Q.qtype_id(bigdl_transformers_low_bit)
        if embedding_qtype is not None:
            embedding_qtype = ggml_tensor_qtype[embedding_qtype]
        elif bigdl_transformers_low_bit in ("gguf_iq2_xxs", "gguf_iq2_xs", "gguf_iq1_s", "q2_k") and not cpu_embedding:
            embedding_qtype = ggml_tensor_qtype["q2_k"]
        elif bigdl_transformers_low_bit in ("gguf_q4k_s", "gguf_q4k_m") and not cpu_embedding:
            embedding_qtype = ggml_tensor_qtype["q4_k"]

        if torch_dtype in ("auto", None):
            dt = getattr(config, "dtype", None) or getattr(config, "torch_dtype", None) or torch.float32
            if isinstance(dt, str):
                dt = getattr(torch, dt, torch.float32)
            if bigdl_transformers_low_bit == "fp16":
                dt = torch.float16
            elif bigdl_transformers_low_bit == "bf16":
                dt = torch.bfloat16
        else:
            dt = torch_dtype
        # 1. skeleton on the meta device
        with torch.device("meta"):
            model = cls.HF_Model.from_config(config, dtype=dt, trust_remote_code=trust_remote_code)
        # 2. shape-only low-bit linears
        model = ggml_convert_low_bit(model, qtype, optimize_model, device="meta", convert_shape_only=True,
                                     modules_to_not_convert=modules_to_not_convert, cpu_embedding=False,
                                     embedding_qtype=embedding_qtype, mixed_precision=mixed_precision,
                                     torch_dtype=dt)
        # 3. fill from the checkpoint files
        f, sharded = extract_local_archive_file(pretrained_model_name_or_path, subfolder, variant)
        files = get_local_shard_files(pretrained_model_name_or_path, f, subfolder)[0] if sharded else [f]
        loaded = {}
        for ck in files:
            sd = load_state_dict(ck)
            _assign_state_dict(model, sd)
            loaded.update({k: None for k in sd})
            del sd
        keys_file = os.path.join(str(pretrained_model_name_or_path), subfolder, "load_keys.json")
        if os.path.isfile(keys_file):
            with open(keys_file) as fk:
                keys_doc = json.load(fk)
            expected = keys_doc.get("all_checkpoint_keys", [])
            from .low_bit_linear import apply_tensor_qtype_overrides
            apply_tensor_qtype_overrides(model, keys_doc.get("tensor_qtypes"))       # mixed formats: per-tensor outcome
            absent = [k for k in expected if k not in loaded]
            invalidInputError(not absent, f"checkpoint is incomplete, missing tensors: {absent[:5]} ...")
        _materialize_meta_buffers(model)
        try:
            head = model.get_output_embeddings()
        except Exception:
            head = None
        if head is None or type(head) is torch.nn.Linear:       # (a low-bit head owns its packed blocks: never re-tied)
            try:
                model.tie_weights()
            except Exception:
                pass
        still_meta = [n for n, p in model.named_parameters() if p.device.type == "meta"]
        if still_meta and getattr(config, "tie_word_embeddings", False) and still_meta == ["lm_head.weight"]:
            emb = model.get_input_embeddings()
            model.lm_head.weight = emb.weight
            still_meta = []
        invalidInputError(not still_meta, f"parameters missing from the checkpoint: {still_meta[:5]}")
        model.eval()
        model.requires_grad_(False)
        if cpu_embedding:
            from .embedding import CPUEmbedding
            for name, module in list(model.named_modules()):
                for cname, child in list(module.named_children()):
                    if type(child) == torch.nn.Embedding:
                        module._modules[cname] = CPUEmbedding.from_embedding(child)
        if disk_embedding or config_dict.get("bigdl_disk_embedding", False):
            from .embedding import DiskEmbedding
            DiskEmbedding.replace_normal_embedding(model)
        gen_cfg = os.path.join(str(pretrained_model_name_or_path), subfolder, "generation_config.json")
        if hasattr(model, "generation_config") and os.path.isfile(gen_cfg):
            try:
                model.generation_config = transformers.GenerationConfig.from_pretrained(
                    pretrained_model_name_or_path, subfolder=subfolder)
            except Exception:
                pass
        # the options this module tree was actually built with (keyword overrides included), so that a save_low_bit of
        # the loaded model describes itself and a later plain load_low_bit rebuilds the same skeleton
        model.config.update({"bigdl_convert_options": {"optimize_model": bool(optimize_model),
                                                       "modules_to_not_convert": list(modules_to_not_convert or []),
                                                       "mixed_precision": bool(mixed_precision)}})
        model.save_low_bit = types.MethodType(save_low_bit, model)
        if pipeline_parallel_stages > 1 or _one_rank_pipeline(pipeline_parallel_stages):
            from .pipeline_parallel import pipeline_parallel, pipeline_parallel_generate
            model = pipeline_parallel(model, pipeline_parallel_stages, dt)
            model.pipeline_parallel_generate = types.MethodType(pipeline_parallel_generate, model)
            torch.distributed.barrier()
        from .lookup import lookup_generate
        model.lookup_generate = types.MethodType(lookup_generate, model)
        if speculative:
            invalidInputError(False, "speculative=True needs the full-precision weights: use from_pretrained")
        return model


class AutoModelForCausalLM(_BaseAutoModelClass):
    HF_Model = transformers.AutoModelForCausalLM


class AutoModel(_BaseAutoModelClass):
    HF_Model = transformers.AutoModel


class AutoModelForSpeechSeq2Seq(_BaseAutoModelClass):
    HF_Model = transformers.AutoModelForSpeechSeq2Seq


class AutoModelForSeq2SeqLM(_BaseAutoModelClass):
    HF_Model = transformers.AutoModelForSeq2SeqLM


class AutoModelForSequenceClassification(_BaseAutoModelClass):
    HF_Model = transformers.AutoModelForSequenceClassification


class AutoModelForMaskedLM(_BaseAutoModelClass):
    HF_Model = transformers.AutoModelForMaskedLM


class AutoModelForQuestionAnswering(_BaseAutoModelClass):
    HF_Model = transformers.AutoModelForQuestionAnswering


class AutoModelForNextSentencePrediction(_BaseAutoModelClass):
    HF_Model = transformers.AutoModelForNextSentencePrediction


class AutoModelForMultipleChoice(_BaseAutoModelClass):
    HF_Model = transformers.AutoModelForMultipleChoice


class AutoModelForTokenClassification(_BaseAutoModelClass):
    HF_Model = transformers.AutoModelForTokenClassification


def _optional_hf(name):
    return getattr(transformers, name, None)


if _optional_hf("Qwen2VLForConditionalGeneration") is not None:
    class Qwen2VLForConditionalGeneration(_BaseAutoModelClass):
        HF_Model = transformers.Qwen2VLForConditionalGeneration

if _optional_hf("AutoModelForImageTextToText") is not None:
    class AutoModelForImageTextToText(_BaseAutoModelClass):
        HF_Model = transformers.AutoModelForImageTextToText

if _optional_hf("AutoModelForVision2Seq") is not None:
    class AutoModelForVision2Seq(_BaseAutoModelClass):
        HF_Model = transformers.AutoModelForVision2Seq
