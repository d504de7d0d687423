"""``ipex_llm.ggml`` namespace: qtype table and the file-level quantizer entry point."""
