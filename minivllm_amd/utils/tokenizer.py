"""Tokenizer access with an offline fallback.

The reference hard-requires ``AutoTokenizer.from_pretrained`` (reference:
src/myvllm/engine/llm_engine.py:34), which needs either network access or a local HF cache.
``load_tokenizer`` tries the HF tokenizer from local files only and otherwise falls back to a
byte-level tokenizer, so the engine plumbing works on an isolated machine.
"""
from __future__ import annotations

from typing import List, Optional


class ByteTokenizer:
    """UTF-8 bytes + a few specials.  ids 0..255 are bytes; ids above are specials that
    decode to nothing.  Token ids >= 259 (random-weight models emit any id) decode to
    the Unicode replacement character."""

    def __init__(self, vocab_size: int = 259):
        self.bos_token_id = 256
        self.eos_token_id = 257
        self.pad_token_id = 258
        self.vocab_size = max(vocab_size, 259)

    def encode(self, text: str, add_special_tokens: bool = False) -> List[int]:
        ids = list(text.encode("utf-8"))
        return ([self.bos_token_id] + ids) if add_special_tokens else (ids or [self.bos_token_id])

    def decode(self, ids, skip_special_tokens: bool = True) -> str:
        out = bytearray()
        for i in ids:
            i = int(i)
            if i < 256:
                out.append(i)
            elif i >= 259:
                out.extend("\ufffd".encode("utf-8"))
        return out.decode("utf-8", errors="replace")

    def apply_chat_template(self, messages, tokenize: bool = False,
                            add_generation_prompt: bool = True, **kw):
        text = "".join(f"<|im_start|>{m['role']}\n{m['content']}<|im_end|>\n" for m in messages)
        if add_generation_prompt:
            text += "<|im_start|>assistant\n"
        return self.encode(text) if tokenize else text

    def __call__(self, text, **kw):
        return {"input_ids": self.encode(text)}


def load_tokenizer(name_or_path: Optional[str], vocab_size: int = 259):
    if name_or_path and name_or_path not in ("byte", "bytes"):
        try:
            from transformers import AutoTokenizer
            return AutoTokenizer.from_pretrained(name_or_path, local_files_only=True)
        except Exception:
            pass
    return ByteTokenizer(vocab_size)
