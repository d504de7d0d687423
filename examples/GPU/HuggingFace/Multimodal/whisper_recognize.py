"""Speech recognition with a low-bit Whisper model
(reference: python/llm/example/GPU/HuggingFace/Multimodal/whisper/recognize.py)."""
import argparse

import torch
from transformers import WhisperProcessor

from bigdl_amd.serving.fastapi.api_server import decode_wav
from ipex_llm.transformers import AutoModelForSpeechSeq2Seq

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", default="openai/whisper-tiny")
    ap.add_argument("--audio", required=True, help="PCM wav file")
    ap.add_argument("--language", default="english")
    a = ap.parse_args()
    model = AutoModelForSpeechSeq2Seq.from_pretrained(a.repo_id_or_model_path, load_in_4bit=True).half().to("cuda")
    model.config.forced_decoder_ids = None
    proc = WhisperProcessor.from_pretrained(a.repo_id_or_model_path)
    forced = proc.get_decoder_prompt_ids(language=a.language, task="transcribe")
    feats = proc(decode_wav(open(a.audio, "rb").read()), sampling_rate=16000, return_tensors="pt").input_features
    with torch.inference_mode():
        ids = model.generate(feats.half().to("cuda"), forced_decoder_ids=forced)
    print(proc.batch_decode(ids, skip_special_tokens=True)[0])
