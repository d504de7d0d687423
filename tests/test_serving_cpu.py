"""Lightweight serving routes against a tiny low-bit model (FastAPI TestClient, CPU)."""
import json

import pytest
import torch
from fastapi.testclient import TestClient
from tokenizers import Tokenizer
from tokenizers.models import WordLevel
from tokenizers.pre_tokenizers import Whitespace
from transformers import LlamaConfig, LlamaForCausalLM, PreTrainedTokenizerFast


@pytest.fixture(scope="module")
def client():
    from bigdl_amd.optimize import optimize_model
    from bigdl_amd.serving.fastapi import FastApp
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=256, tie_word_embeddings=False)
    model = LlamaForCausalLM(cfg).eval()
    model.generation_config.eos_token_id = None
    model = optimize_model(model, low_bit="sym_int8")
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    vocab.update({f"w{i}": i + 3 for i in range(317)})
    t = Tokenizer(WordLevel(vocab, unk_token="<unk>"))
    t.pre_tokenizer = Whitespace()
    tok = PreTrainedTokenizerFast(tokenizer_object=t, bos_token="<s>", eos_token="</s>", unk_token="<unk>")
    fast = FastApp(model, tok, served_model_name="tiny")
    with TestClient(fast.app) as c:
        c.model, c.tok = model, tok
        yield c


def _greedy_text(client, prompt, n):
    ids = client.tok(prompt, return_tensors="pt").input_ids
    out = client.model.generate(ids, max_new_tokens=n, do_sample=False)
    return client.tok.decode(out[0, ids.shape[1]:], skip_special_tokens=True)


def test_generate_and_stream_match_direct_generation(client):
    ref = _greedy_text(client, "w1 w2 w3", 6)
    r = client.post("/generate", json={"inputs": "w1 w2 w3", "parameters": {"max_new_tokens": 6}})
    assert r.status_code == 200 and r.json()["generated_text"] == ref and r.json()["generate_tokens"] == 6
    with client.stream("POST", "/generate_stream", json={"inputs": "w1 w2 w3", "parameters": {"max_new_tokens": 6}}) as s:
        chunks = [l[len("data: "):] for l in s.iter_lines() if l.startswith("data: ")]
    assert chunks[-1] == "[DONE]"
    text = "".join(json.loads(c)["token"]["text"] for c in chunks[:-1])
    assert text == ref


def test_openai_routes(client):
    ref = _greedy_text(client, "w5 w6", 5)
    r = client.post("/v1/completions", json={"model": "tiny", "prompt": "w5 w6", "max_tokens": 5})
    body = r.json()
    assert r.status_code == 200 and body["choices"][0]["text"] == ref
    assert body["usage"]["prompt_tokens"] == 2 and body["usage"]["completion_tokens"] == 5
    r = client.post("/v1/chat/completions", json={"model": "tiny", "max_tokens": 4,
                                                  "messages": [{"role": "user", "content": "w7 w8"}]})
    assert r.status_code == 200 and r.json()["choices"][0]["message"]["role"] == "assistant"
    with client.stream("POST", "/v1/chat/completions", json={"model": "tiny", "max_tokens": 4, "stream": True,
                                                             "messages": [{"role": "user", "content": "w7 w8"}]}) as s:
        lines = [l[len("data: "):] for l in s.iter_lines() if l.startswith("data: ")]
    assert lines[-1] == "[DONE]" and json.loads(lines[0])["choices"][0]["delta"]["role"] == "assistant"
    streamed = "".join(json.loads(l)["choices"][0]["delta"].get("content") or "" for l in lines[1:-1])
    assert streamed == r.json()["choices"][0]["message"]["content"]
    assert client.get("/v1/models").json()["data"][0]["id"] == "tiny"


def test_stop_strings_health_and_metrics(client):
    ref = _greedy_text(client, "w9 w10", 8)
    words = ref.split()
    stop = words[2]
    r = client.post("/v1/completions", json={"prompt": "w9 w10", "max_tokens": 8, "stop": [stop]})
    got = r.json()["choices"][0]["text"]
    assert stop not in got and ref.startswith(got)
    assert client.get("/health").json()["status"] == "ok"
    m = client.get("/metrics").text
    assert "bigdl_requests_total" in m and "bigdl_generated_tokens_total" in m


def test_audio_transcription_route():
    """/v1/audio/transcriptions: multipart WAV upload -> Whisper-style generate -> text (tiny random model, stub
    processor: the route, the stdlib multipart / WAV decoding and the dtype / device plumbing are what is tested)."""
    import io
    import wave

    import numpy as np
    import torch
    from starlette.testclient import TestClient
    from transformers import WhisperConfig, WhisperForConditionalGeneration

    from bigdl_amd.optimize import optimize_model
    from bigdl_amd.serving.fastapi.api_server import FastApp, decode_wav, parse_multipart

    torch.manual_seed(0)
    cfg = WhisperConfig(vocab_size=320, d_model=128, encoder_layers=1, decoder_layers=1, encoder_attention_heads=4,
                        decoder_attention_heads=4, encoder_ffn_dim=256, decoder_ffn_dim=256, num_mel_bins=16,
                        max_source_positions=32, max_target_positions=32, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                        decoder_start_token_id=1)
    model = optimize_model(WhisperForConditionalGeneration(cfg).eval(), low_bit="sym_int8")
    model.generation_config.forced_decoder_ids = None

    class Proc:
        seen = None

        def __call__(self, audio, sampling_rate=16000, return_tensors="pt"):
            Proc.seen = (len(audio), sampling_rate)
            g = torch.Generator().manual_seed(len(audio))

            class R:
                input_features = torch.randn(1, 16, 64, generator=g)
            return R

        def batch_decode(self, ids, skip_special_tokens=True):
            return ["tokens:" + ",".join(str(int(t)) for t in ids[0])]

    class Tok:
        eos_token_id = 2

    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(8000)
        t = np.arange(8000) / 8000.0
        pcm = (np.sin(2 * np.pi * 440 * t) * 12000).astype(np.int16)
        w.writeframes(np.stack([pcm, pcm], 1).tobytes())
    wav = buf.getvalue()
    assert decode_wav(wav).shape == (16000,)                      # stereo 8 kHz -> mono 16 kHz
    app = FastApp(model=model, mytokenizer=Tok(), myprocessor=Proc())
    with TestClient(app.app) as c:
        r = c.post("/v1/audio/transcriptions", files={"file": ("a.wav", wav, "audio/wav")},
                   data={"language": "", "response_format": "json"})
        assert r.status_code == 200, r.text
        assert r.json()["text"].startswith("tokens:1") and Proc.seen == (16000, 16000)
        bad = c.post("/v1/audio/transcriptions", files={"file": ("a.wav", b"not a wav", "audio/wav")})
        assert bad.status_code == 400
    form = parse_multipart('multipart/form-data; boundary=XX',
                           b'--XX\r\nContent-Disposition: form-data; name="language"\r\n\r\nen\r\n--XX--\r\n')
    assert form == {"language": "en"}
