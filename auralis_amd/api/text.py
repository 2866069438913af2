"""Minimal text front-end feeding the hot path: sentence chunking and BPE ids.

Reference: XTTSTokenizerFast / split_sentence / char_limits (src/auralis/models/xttsv2/config/tokenizer.py:119-236,
742-1002).  Scope note: text normalisation lives in api/cleaners.py (en/fr/de numbers spelled out; other languages
keep digits); the spaCy sentencizer is replaced by a punctuation sentencizer; this module keeps the contract the engine
depends on — chunks no longer than the per-language character limit,
ids = BPE("[lang]" + text with " " -> "[SPACE]") wrapped in [START]/[STOP] — and uses the real tokenizer.json when
the checkpoint directory has one, else a deterministic stand-in vocabulary for synthetic checkpoints."""
from __future__ import annotations

import os
import re
import zlib
from typing import List, Optional

CHAR_LIMITS = {"en": 250, "de": 253, "fr": 273, "es": 239, "it": 213, "pt": 203, "pl": 224, "zh": 82, "ar": 166,
               "cs": 186, "ru": 182, "nl": 251, "tr": 226, "ja": 71, "hu": 224, "ko": 95}
_SENT_END = re.compile(r"(?<=[.!?;:。！？])\s+|\n{2,}")
_SOFT = re.compile(r"[,)\]\-–—]\s|\s")


def _best_split(text: str, limit: int, window: int = 30) -> int:
    """Split position <= limit, preferring punctuation then whitespace inside the last `window` characters."""
    lo = max(1, limit - window)
    seg = text[lo:limit]
    for pat in (r"[.!?;:]\s", r"[,)\]]\s", r"[-–—]\s", r"\s"):
        hits = list(re.finditer(pat, seg))
        if hits:
            return lo + hits[-1].end()
    return limit


def split_sentence(text: str, lang: str, text_split_length: int = 250) -> List[str]:
    text = text.strip()
    if len(text) <= text_split_length:
        return [text]
    sentences = [s.strip() for s in _SENT_END.split(text) if s and s.strip()]
    chunks: List[str] = []
    cur: List[str] = []
    cur_len = 0
    for s in sentences:
        if cur_len + len(s) <= text_split_length:
            cur.append(s)
            cur_len += len(s) + 1
            continue
        if cur:
            chunks.append(" ".join(cur))
            cur, cur_len = [], 0
        while len(s) > text_split_length:
            p = _best_split(s, text_split_length)
            chunks.append(s[:p].strip())
            s = s[p:].strip()
        if s:
            cur, cur_len = [s], len(s)
    if cur:
        chunks.append(" ".join(cur))
    # the reference replaces a trailing '.' by a space ("prevents annoying sounds", tokenizer.py:234)
    return [c[:-1] + " " if c.endswith(".") else c for c in chunks if c]


class XTTSTokenizer:
    """ids for one chunk.  bos/eos = [START]/[STOP]; synthetic stand-in uses ids 261/0 like the survey's fixture."""

    def __init__(self, tokenizer_file: Optional[str] = None, vocab_size: int = 6681):
        self.vocab_size = vocab_size
        self._tok = None
        if tokenizer_file and os.path.isfile(tokenizer_file):
            from tokenizers import Tokenizer
            self._tok = Tokenizer.from_file(tokenizer_file)
            self.bos_token_id = self._tok.token_to_id("[START]")
            self.eos_token_id = self._tok.token_to_id("[STOP]")
        else:
            self.bos_token_id, self.eos_token_id = 261, 0

    def char_limit(self, lang: str) -> int:
        return CHAR_LIMITS.get(lang.split("-")[0], 250)

    def encode_chunk(self, text: str, lang: str) -> List[int]:
        from .cleaners import multilingual_cleaners
        base = lang.split("-")[0]
        code = "zh-cn" if base == "zh" else base
        clean = multilingual_cleaners(text, lang)           # tokenizer.py:708-719 order
        s = f"[{code}]{clean}".replace(" ", "[SPACE]")
        if self._tok is not None:
            ids = self._tok.encode(s, add_special_tokens=False).ids
        else:
            # stand-in: ~3 characters per token, stable across runs/platforms
            raw = f"[{code}]{clean}"
            ids = [2 + zlib.crc32(raw[i:i + 3].encode("utf-8")) % (self.vocab_size - 300) for i in range(0, len(raw), 3)]
            ids = [i if i != 261 else 262 for i in ids]
        return [self.bos_token_id] + ids + [self.eos_token_id]

    def batch_encode_with_split(self, text: str, lang: str) -> List[List[int]]:
        return [self.encode_chunk(c, lang) for c in split_sentence(text, lang, self.char_limit(lang))]
