"""text-generation-inference style schemas for ``/generate`` and ``/generate_stream``
(reference: python/llm/src/ipex_llm/serving/fastapi/tgi_protocol.py)."""
from __future__ import annotations

from typing import List, Optional

from pydantic import BaseModel


class Parameters(BaseModel):
    max_new_tokens: int = 32
    min_new_tokens: Optional[int] = None
    do_sample: Optional[bool] = None
    temperature: Optional[float] = None
    top_p: Optional[float] = None
    top_k: Optional[int] = None
    repetition_penalty: Optional[float] = None
    stop: Optional[List[str]] = None


class InputsRequest(BaseModel):
    inputs: str
    parameters: Optional[Parameters] = None
    image_list: Optional[list] = None
    stream: Optional[bool] = False
    req_type: str = "completion"


class GenerateResponse(BaseModel):
    generated_text: str
    generate_tokens: int = 0
    finish_reason: Optional[str] = None


class Token(BaseModel):
    text: str


class StreamResponse(BaseModel):
    token: Token
    generated_text: Optional[str] = None
    details: Optional[dict] = None
