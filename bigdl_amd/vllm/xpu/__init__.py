"""Import path of the reference (ipex_llm.vllm.xpu); on MI355X there is one implementation: bigdl_amd.vllm."""
