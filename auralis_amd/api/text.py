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


def preprocess_text(text: str, lang: str) -> str:
    """XTTSTokenizerFast.preprocess_text (tokenizer.py:805-820): multilingual cleaners for the 15 alphabetic tags, then
    romanisation for zh (pypinyin TONE3, tokenizer.py:727-730) and ko (hangul_romanize academic, :737-738); ja goes through
    cutlet romaji + lowercase (:732-735); every other tag through basic_cleaners (lowercase + collapse whitespace, :721-725).
    The BPE vocabulary is built on ROMANISED text: without the transliteration package the ids would be garbage, so a
    missing package is an error, never a silent pass-through."""
    from .cleaners import multilingual_cleaners
    base = lang.split("-")[0]
    if base in {"ar", "cs", "de", "en", "es", "fr", "hu", "it", "nl", "pl", "pt", "ru", "tr", "zh", "ko"}:
        text = multilingual_cleaners(text, base)
        if base == "zh":
            try:
                import pypinyin
            except ImportError as e:
                raise NotImplementedError("language 'zh-cn' needs the pypinyin package (pinyin transliteration before BPE, as "
                                          "the reference does)") from e
            text = "".join(p[0] for p in pypinyin.pinyin(text, style=pypinyin.Style.TONE3, heteronym=False,
                                                          neutral_tone_with_five=True))
        if base == "ko":
            try:
                from hangul_romanize import Transliter
                from hangul_romanize.rule import academic
            except ImportError as e:
                raise NotImplementedError("language 'ko' needs the hangul_romanize package (romanisation before BPE, as the "
                                          "reference does)") from e
            text = Transliter(academic).translit(text)
        return text
    if base == "ja":
        try:
            import cutlet
        except ImportError as e:
            raise NotImplementedError("language 'ja' needs the cutlet package (romaji before BPE, as the reference does)") from e
        return cutlet.Cutlet().romaji(text).lower()
    return re.sub(r"\s+", " ", text.lower())


class XTTSTokenizer:
    """ids for one chunk.  bos/eos = [START]/[STOP].

    A real checkpoint needs its tokenizer.json (the reference loads it from the gpt_model repo): without one the constructor
    raises.  `synthetic=True` selects a deterministic stand-in vocabulary (ids 261/0 as bos/eos like the survey's fixture)
    for the seeded synthetic checkpoints of the tests and the bench; it is never chosen implicitly."""

    def __init__(self, tokenizer_file: Optional[str] = None, vocab_size: int = 6681, synthetic: bool = False):
        self.vocab_size = vocab_size
        self._tok = None
        if tokenizer_file and os.path.isfile(tokenizer_file):
            from tokenizers import Tokenizer
            self._tok = Tokenizer.from_file(tokenizer_file)
            self.bos_token_id = self._tok.token_to_id("[START]")
            self.eos_token_id = self._tok.token_to_id("[STOP]")
            if self.bos_token_id is None or self.eos_token_id is None:
                raise ValueError(f"{tokenizer_file}: no [START] / [STOP] tokens (not an XTTS tokenizer.json)")
        elif synthetic:
            self.bos_token_id, self.eos_token_id = 261, 0
        else:
            raise FileNotFoundError(
                f"tokenizer.json not found ({tokenizer_file!r}): text ids for a real checkpoint need the XTTS BPE vocabulary "
                "(AstraMindAI/xtts2-gpt ships it next to gpt2_model.safetensors); pass synthetic=True only for the seeded "
                "synthetic checkpoints")

    def char_limit(self, lang: str) -> int:
        return CHAR_LIMITS.get(lang.split("-")[0], 250)

    def encode_chunk(self, text: str, lang: str) -> List[int]:
        base = lang.split("-")[0]
        code = "zh-cn" if base == "zh" else base
        clean = preprocess_text(text, lang)                 # tokenizer.py:805-820
        s = f"[{code}]{clean}".replace(" ", "[SPACE]")      # tokenizer.py:914-917
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
