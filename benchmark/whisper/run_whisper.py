"""Whisper accuracy (word error rate) and speed (real-time factor) with low-bit weights.

Reference: python/llm/dev/benchmark/whisper/run_whisper.py:30-97 (LibriSpeech test-clean / test-other through
``datasets`` + the ``wer`` metric script; reports WER, RTF = processing time / audio time, RTX = 1 / RTF).  Without
network access the data comes from a manifest: one JSON object per line, ``{"audio": "<path to PCM wav>", "text": "..."}``.
WER is computed here (word-level Levenshtein distance over the normalized transcripts)."""
from __future__ import annotations

import argparse
import csv
import json
import os
import re
import time
from typing import List, Sequence


def normalize(text: str) -> List[str]:
    return re.sub(r"[^a-z0-9' ]+", " ", text.lower()).split()


def edit_distance(ref: Sequence[str], hyp: Sequence[str]) -> int:
    prev = list(range(len(hyp) + 1))
    for i, r in enumerate(ref, 1):
        cur = [i] + [0] * len(hyp)
        for j, h in enumerate(hyp, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (r != h))
        prev = cur
    return prev[-1]


def wer(references: Sequence[str], predictions: Sequence[str]) -> float:
    """Corpus WER: total word edits / total reference words."""
    errs = words = 0
    for r, p in zip(references, predictions):
        rw, pw = normalize(r), normalize(p)
        errs += edit_distance(rw, pw)
        words += len(rw)
    return errs / max(1, words)


def main():
    import torch
    ap = argparse.ArgumentParser(description="Evaluate Whisper speed and accuracy")
    ap.add_argument("--model_path", required=True)
    ap.add_argument("--manifest", required=True, help="jsonl with audio (wav path) and text")
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--load_in_low_bit", default="sym_int4")
    ap.add_argument("--limit", type=int, default=500)
    ap.add_argument("--save_result", action="store_true")
    a = ap.parse_args()
    from transformers import WhisperProcessor
    from bigdl_amd.serving.fastapi.api_server import decode_wav
    from ipex_llm.transformers import AutoModelForSpeechSeq2Seq
    proc = WhisperProcessor.from_pretrained(a.model_path)
    forced = proc.get_decoder_prompt_ids(language="en", task="transcribe")
    model = AutoModelForSpeechSeq2Seq.from_pretrained(a.model_path, load_in_low_bit=a.load_in_low_bit, optimize_model=True).eval()
    if a.device != "cpu":
        model = model.half().to(a.device)
    model.config.forced_decoder_ids = None
    dtype = torch.float16 if a.device != "cpu" else torch.float32
    refs, hyps, audio_s, proc_s = [], [], 0.0, 0.0
    rows = [json.loads(l) for l in open(a.manifest) if l.strip()][: a.limit]
    for i, row in enumerate(rows):
        audio = decode_wav(open(row["audio"], "rb").read())
        t0 = time.time()
        feats = proc(audio, sampling_rate=16000, return_tensors="pt").input_features.to(a.device, dtype)
        with torch.no_grad():
            ids = model.generate(feats, forced_decoder_ids=forced, use_cache=True)[0]
        if a.device != "cpu":
            torch.cuda.synchronize()
        dt = time.time() - t0
        refs.append(row["text"])
        hyps.append(proc.decode(ids, skip_special_tokens=True))
        if i > 0:                                   # the first sample pays for warm-up, as in the reference
            audio_s += len(audio) / 16000.0
            proc_s += dt
    w = 100 * wer(refs, hyps)
    rtf = proc_s / max(audio_s, 1e-9)
    print(json.dumps({"model": a.model_path, "precision": a.load_in_low_bit, "WER": round(w, 3), "RTF": round(rtf, 4),
                      "RTX": round(1 / max(rtf, 1e-9), 2), "n": len(rows)}))
    if a.save_result:
        os.makedirs("results", exist_ok=True)
        name = f"results/{os.path.basename(a.model_path.rstrip('/'))}-{a.device}-{a.load_in_low_bit}.csv"
        new = not os.path.exists(name)
        with open(name, "a", newline="") as f:
            cw = csv.writer(f)
            if new:
                cw.writerow(["models", "precision", "WER", "RTF"])
            cw.writerow([a.model_path, a.load_in_low_bit, w, rtf])


if __name__ == "__main__":
    main()
