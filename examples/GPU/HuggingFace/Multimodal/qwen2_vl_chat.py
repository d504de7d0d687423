"""Image + text chat with Qwen2-VL (reference: python/llm/example/GPU/HuggingFace/Multimodal/qwen2-vl/generate.py).
The vision tower and the merger run as HuggingFace modules (MFMA flash attention through the attention registry, 80-channel
heads included); the text decoder runs on the native engine, which takes the merged embeddings and the 3-axis rotary
positions from the wrapper.

    python qwen2_vl_chat.py --repo-id-or-model-path Qwen/Qwen2-VL-7B-Instruct --image photo.jpg --prompt "Describe the picture."
"""
import argparse
import time

import torch
from transformers import AutoProcessor

from ipex_llm.transformers import Qwen2VLForConditionalGeneration

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--image", required=True)
    ap.add_argument("--prompt", default="Describe this image.")
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--n-predict", type=int, default=128)
    a = ap.parse_args()
    from PIL import Image
    # the vision tower stays in fp16 (it is a small part of the weights and sensitive to quantization)
    model = Qwen2VLForConditionalGeneration.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit,
                                                            optimize_model=True, modules_to_not_convert=["visual"],
                                                            use_cache=True)
    model = model.half().to("cuda")
    processor = AutoProcessor.from_pretrained(a.repo_id_or_model_path)
    messages = [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": a.prompt}]}]
    text = processor.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
    inputs = processor(text=[text], images=[Image.open(a.image).convert("RGB")], return_tensors="pt").to("cuda")
    with torch.inference_mode():
        model.generate(**inputs, max_new_tokens=8)                      # warm-up: engine build + graph capture
        t0 = time.perf_counter()
        out = model.generate(**inputs, max_new_tokens=a.n_predict, do_sample=False)
        torch.cuda.synchronize()
    print(processor.batch_decode(out[:, inputs.input_ids.shape[1]:], skip_special_tokens=True)[0])
    print(f"-- {time.perf_counter() - t0:.2f} s, first token {getattr(model, 'first_cost', 0) * 1e3:.1f} ms, "
          f"{getattr(model, 'rest_cost_mean', 0) * 1e3:.2f} ms/token after it")
