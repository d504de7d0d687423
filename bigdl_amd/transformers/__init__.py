"""``ipex_llm.transformers`` compatible namespace (reference: python/llm/src/ipex_llm/transformers/__init__.py)."""
_LAZY = {
    "AutoModelForCausalLM": "model", "AutoModel": "model", "AutoModelForSeq2SeqLM": "model",
    "AutoModelForSpeechSeq2Seq": "model", "AutoModelForQuestionAnswering": "model",
    "AutoModelForSequenceClassification": "model", "AutoModelForMaskedLM": "model",
    "AutoModelForNextSentencePrediction": "model", "AutoModelForMultipleChoice": "model",
    "AutoModelForTokenClassification": "model", "Qwen2VLForConditionalGeneration": "model",
    "AutoModelForImageTextToText": "model", "AutoModelForVision2Seq": "model",
    "ggml_convert_low_bit": "convert", "get_enable_ipex": "convert",
    "init_pipeline_parallel": "pipeline_parallel", "PPModelWorker": "pipeline_parallel",
    "LlamaForCausalLM": "modelling_bigdl", "ChatGLMForCausalLM": "modelling_bigdl",
    "GptneoxForCausalLM": "modelling_bigdl", "BloomForCausalLM": "modelling_bigdl",
    "StarcoderForCausalLM": "modelling_bigdl", "BigdlNativeForCausalLM": "modelling_bigdl",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod = importlib.import_module(f"{__name__}.{_LAZY[name]}")
        return getattr(mod, name)
    raise AttributeError(name)
