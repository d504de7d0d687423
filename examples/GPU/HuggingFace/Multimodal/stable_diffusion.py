"""Stable Diffusion / SDXL / LCM-LoRA with the fused attention processor (counterpart of the reference's
example/GPU/HuggingFace/Multimodal/StableDiffusion/{openjourney,sdxl,lora-lcm}.py).  Needs ``diffusers`` and a local
checkpoint directory.

    python stable_diffusion.py --repo-id-or-model-path /models/stable-diffusion-xl-base-1.0 --prompt "an astronaut" \
        [--lora-weights-path /models/lcm-lora-sdxl] [--low-bit sym_int8]
"""
import argparse
import time

import torch

from bigdl_amd import optimize_model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt", default="An astronaut in the forest, detailed, 8k")
    ap.add_argument("--lora-weights-path", default=None, help="LCM-LoRA weights: 4 steps, guidance 1.0")
    ap.add_argument("--low-bit", default=None, help="also quantize the denoiser's linears (e.g. sym_int8)")
    ap.add_argument("--num-steps", type=int, default=20)
    ap.add_argument("--save-path", default="sd-output.png")
    a = ap.parse_args()
    from diffusers import DiffusionPipeline
    pipe = DiffusionPipeline.from_pretrained(a.repo_id_or_model_path, torch_dtype=torch.float16, use_safetensors=True)
    steps, extra = a.num_steps, {}
    if a.lora_weights_path:
        from diffusers import LCMScheduler
        pipe.scheduler = LCMScheduler.from_config(pipe.scheduler.config)
        pipe.load_lora_weights(a.lora_weights_path)
        steps, extra = 4, {"guidance_scale": 1.0}
    pipe = optimize_model(pipe, low_bit=a.low_bit)          # fused attention processor (+ bf16 VAE guard on SDXL)
    pipe.to("cuda")
    with torch.inference_mode():
        pipe(prompt=a.prompt, num_inference_steps=steps, **extra)                     # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        image = pipe(prompt=a.prompt, num_inference_steps=steps, **extra).images[0]
        torch.cuda.synchronize()
    print(f"{steps} steps in {time.perf_counter() - t0:.2f} s")
    image.save(a.save_path)


if __name__ == "__main__":
    main()
