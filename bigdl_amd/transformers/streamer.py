"""Text streamers (reference: python/llm/src/ipex_llm/transformers/streamer.py:25-110 ``BatchTextIteratorStreamer``).

``BatchTextIteratorStreamer`` streams ``batch_size`` sequences at once: each ``put`` of a [B] / [B, 1] token tensor
yields a list of B text deltas on the iterator side.
"""
from __future__ import annotations

from queue import Queue
from typing import Optional

import torch
from transformers import TextIteratorStreamer


class BatchTextIteratorStreamer(TextIteratorStreamer):
    def __init__(self, batch_size: int, tokenizer, skip_prompt: bool = False, timeout: Optional[float] = None,
                 **decode_kwargs):
        super().__init__(tokenizer, skip_prompt, timeout, **decode_kwargs)
        self.batch_size = batch_size
        self.token_cache = [[] for _ in range(batch_size)]
        self.print_len = [0] * batch_size
        self.generate_exception = None
        self.text_queue = Queue()

    def put(self, value):
        if isinstance(value, torch.Tensor):
            value = value.reshape(self.batch_size, -1) if value.numel() >= self.batch_size else value.reshape(1, -1)
        if self.skip_prompt and self.next_tokens_are_prompt:
            self.next_tokens_are_prompt = False
            return
        out = []
        for i in range(self.batch_size):
            self.token_cache[i].extend(value[i].tolist())
            text = self.tokenizer.decode(self.token_cache[i], **self.decode_kwargs)
            if text.endswith("\n"):
                printable = text[self.print_len[i]:]
                self.token_cache[i] = []
                self.print_len[i] = 0
            elif len(text) > 0 and _is_cjk(ord(text[-1])):
                printable = text[self.print_len[i]:]
                self.print_len[i] += len(printable)
            else:
                # hold back the last partial word so multi-token characters are not split
                printable = text[self.print_len[i]: text.rfind(" ") + 1]
                self.print_len[i] += len(printable)
            out.append(printable)
        self.on_finalized_text(out)

    def end(self):
        out = []
        for i in range(self.batch_size):
            if self.token_cache[i]:
                text = self.tokenizer.decode(self.token_cache[i], **self.decode_kwargs)
                out.append(text[self.print_len[i]:])
                self.token_cache[i] = []
                self.print_len[i] = 0
            else:
                out.append("")
        self.next_tokens_are_prompt = True
        self.on_finalized_text(out, stream_end=True)

    def on_finalized_text(self, texts, stream_end: bool = False):
        self.text_queue.put(texts, timeout=self.timeout)
        if stream_end:
            self.text_queue.put(self.stop_signal, timeout=self.timeout)


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF
            or 0x2F800 <= cp <= 0x2FA1F)
