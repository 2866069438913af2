"""Text front-end feeding the hot path: sentence chunking and BPE ids.

Reference: XTTSTokenizerFast / split_sentence / find_best_split_point / char_limits
(src/auralis/models/xttsv2/config/tokenizer.py:51-236, 742-1002).  split_sentence follows the reference's algorithm step by
step; spaCy (absent offline) contributes only its rule-based `sentencizer`, restated in _sentencize with the tokenizer
approximated on whitespace words (abbreviation exceptions listed there).  Text normalisation lives in api/cleaners.py (en/fr/de
numbers spelled out; other languages keep digits).  ids = BPE("[lang]" + cleaned text with " " -> "[SPACE]") wrapped in
[START]/[STOP], from the checkpoint's tokenizer.json (required for real checkpoints)."""
from __future__ import annotations

import os
import re
import zlib
from typing import List, Optional

CHAR_LIMITS = {"en": 250, "de": 253, "fr": 273, "es": 239, "it": 213, "pt": 203, "pl": 224, "zh": 82, "ar": 166,
               "cs": 186, "ru": 182, "nl": 251, "tr": 226, "ja": 71, "hu": 224, "ko": 95}
# spaCy `sentencizer` default punct_chars (the subset that occurs in the scripts XTTS supports): a token made of these ends a
# sentence; following punctuation-only tokens (closing quotes / brackets) still belong to it
_SENT_PUNCT = ".!?։؟۔܀܁܂߹।॥၊။።፧፨᙮᜵᜶᠃᠉᥄᥅᪨᪩᪪᪫᭚᭛᭞᭟᰻᰼᱾᱿‼‽⁇⁈⁉⸮⸼꓿꘎꘏꛳꛷꡶꡷꣎꣏꤯꧈꧉꩝꩞꩟꫰꫱꯫﹒﹖﹗！．？｡。"
_CLOSERS = "\"'”’»›)]}）】』》"
# spaCy's English tokenizer keeps these as ONE token (tokenizer exceptions), so their period does not end a sentence
_ABBREV = {"mr.", "mrs.", "ms.", "dr.", "prof.", "st.", "sr.", "jr.", "mt.", "vs.", "etc.", "e.g.", "i.e.", "a.m.", "p.m.",
           "no.", "co.", "inc.", "ltd.", "corp.", "gen.", "gov.", "sen.", "rep.", "adm.", "messrs.", "bros.", "jan.", "feb.",
           "mar.", "apr.", "jun.", "jul.", "aug.", "sep.", "sept.", "oct.", "nov.", "dec.", "ph.d.", "u.s.", "u.k."}


def _sentencize(text: str, lang: str = "en") -> List[str]:
    """Rule-based sentence boundaries as spaCy's `sentencizer` draws them on a blank pipeline (tokenizer.py:185-191): a
    sentence ends after a token consisting of sentence punctuation (plus any punctuation-only tokens after it) and the next
    sentence starts at the following token.  The tokenizer is approximated on whitespace-separated words: a word's trailing run
    of sentence punctuation (optionally followed by closing quotes / brackets) is that punctuation token, unless the word is
    one of the tokenizer's abbreviation exceptions, a single letter + '.', or a number / dotted token with no space (3.14,
    example.com keep their dots because the period is word-internal).  zh / ja have no spaces: spaCy's Chinese() tokenizer
    is character-level there, so every sentence-punctuation character (plus closers after it) ends a sentence wherever it
    stands in the string."""
    if lang in ("zh", "ja"):
        out, start, i, n = [], 0, 0, len(text)
        while i < n:
            if text[i] in _SENT_PUNCT:
                j = i + 1
                while j < n and (text[j] in _SENT_PUNCT or text[j] in _CLOSERS):
                    j += 1
                out.append(text[start:j])
                start = i = j
            else:
                i += 1
        if text[start:].strip():
            out.append(text[start:])
        return out
    out, start = [], 0
    for m in re.finditer(r"\S+", text):
        w = m.group(0)
        core = w.rstrip(_CLOSERS)
        if not core or core[-1] not in _SENT_PUNCT:
            continue
        if core[-1] == ".":
            low = core.lower().lstrip("\"'“‘«‹([{")
            if low in _ABBREV or re.fullmatch(r"[a-z]\.", low) or re.fullmatch(r"(?:[a-z]\.){2,}", low):
                continue
        out.append(text[start:m.end()])
        start = m.end()
    if text[start:].strip():
        out.append(text[start:])
    return out


def find_best_split_point(text: str, target_pos: int, window_size: int = 30) -> int:
    """tokenizer.py:51-115 restated: every break marker inside [target - window, target + window] is scored
    priority x (1 - distance / (2 * window)) and the best one wins (first best on ties); the split lands AFTER the marker.
    Note the window extends past the target, so a chunk may exceed the limit by up to `window_size` characters, as in the
    reference."""
    markers = ((r"[.!?؟။။။]+[\s]*", 1.0), (r"[\n\r]+\s*[\n\r]+", 1.0), (r"[:|;；：；][\s]*", 0.9), (r"[,，،、][\s]*", 0.8),
               (r"[)}\]）】』»›》\s]+", 0.7), (r"[-—−]+[\s]*", 0.7), (r"\s+[&+=/\s]+\s+", 0.6), (r"[\s]+", 0.5))
    start = max(0, target_pos - window_size)
    window = text[start:min(len(text), target_pos + window_size)]
    best_pos, best_score = target_pos, 0.0
    for pattern, priority in markers:
        for m in re.finditer(pattern, window):
            pos = start + m.end()
            score = priority * (1 - abs(pos - target_pos) / (window_size * 2))
            if score > best_score:
                best_score, best_pos = score, pos
    return best_pos


def split_sentence(text: str, lang: str, text_split_length: int = 250, sentences: Optional[List[str]] = None) -> List[str]:
    """tokenizer.py:119-236: sentences (spaCy sentencizer, restated in _sentencize; `sentences` injects a precomputed list, as
    the golden test does) are packed greedily into chunks of at most `text_split_length` characters; a sentence longer than
    the limit is cut at find_best_split_point; a trailing '.' of a chunk becomes a space (":234, prevents annoying sounds")."""
    text = text.strip()
    if len(text) <= text_split_length:
        return [text]
    splits: List[str] = []
    cur: List[str] = []
    cur_len = 0
    for sent in (sentences if sentences is not None else _sentencize(text, lang.split("-")[0])):
        s = sent.strip()
        n = len(s)
        if cur_len + n <= text_split_length:
            cur.append(s)
            cur_len += n + 1
        elif n > text_split_length:
            if cur:
                splits.append(" ".join(cur))
                cur, cur_len = [], 0
            rest = s
            while len(rest) > text_split_length:
                p = find_best_split_point(rest, text_split_length, window_size=30)
                splits.append(rest[:p].strip())
                rest = rest[p:].strip()
            if rest:
                cur, cur_len = [rest], len(rest)
        else:
            splits.append(" ".join(cur))
            cur, cur_len = [s], n
    if cur:
        splits.append(" ".join(cur))
    return [c[:-1] + " " if c.endswith(".") else c for c in splits if c]


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
