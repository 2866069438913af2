"""Run the reference's own text front-end (src/auralis/models/xttsv2/config/tokenizer.py) UNMODIFIED from /root/reference.

TEST INFRASTRUCTURE ONLY (build container; /root/reference does not exist on the GPU box).  The module imports spacy,
num2words, pypinyin, cutlet and hangul_romanize at module level; none is installed here.  They are satisfied by stub
modules, so everything the reference *defines* in that file — find_best_split_point, split_sentence's packing loop, the
abbreviation / symbol tables and their expanders, the number regexes and their order, multilingual_cleaners,
XTTSTokenizerFast.preprocess_text — runs as written.  What the stubs replace, and what a fixture made through them can
therefore NOT pin:

* spaCy's tokenizer + `sentencizer`: `FakeNLP` returns the sentence list injected with `set_sentences()`; the fixture
  stores that list, so the packing loop is pinned while the sentence boundaries themselves are not;
* `num2words`: `marker_num2words` returns a structural, digit-free marker ("<card bc en>", "<cur bc USD en>, <cents fa>" ...) so the
  regex plumbing around it (separator removal, currency / decimal / ordinal / cardinal order, the integer-amount tail drop) is
  pinned while the spelling of a number is not;
* pypinyin / cutlet / hangul_romanize (zh / ja / ko romanisation): not exercised.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("AURALIS_REFERENCE", "/root/reference")
_TOK = "src/auralis/models/xttsv2/config/tokenizer.py"
_ZH = "src/auralis/models/xttsv2/components/tts/layers/xtts/zh_num2words.py"

# joiner between the major and the minor unit in num2words' currency strings, per language (the reference's own table,
# tokenizer.py:651-666, is what strips the tail of integer amounts)
_JOIN = {"en": ", ", "es": " con ", "fr": " et ", "de": " und ", "pt": " e ", "it": " e "}


def reference_text_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, _TOK))


def _letters(x) -> str:
    """digits -> letters (0 = a ... 9 = j, '.' = p): real num2words output holds no digits, so the regexes that run after a
    substitution must not find any in the marker either"""
    return "".join("abcdefghij"[int(c)] if c.isdigit() else "p" for c in str(x))


def marker_num2words(value, lang="en", to="cardinal", ordinal=False, currency=None, **_):
    """Structural stand-in for num2words: the call's arguments, spelled as a digit-free marker."""
    if to == "currency":
        major = int(value)
        cents = int(round((value - major) * 100))
        return f"<cur {_letters(major)} {currency} {lang}>{_JOIN.get(lang, ', ')}<cents {_letters(cents)}>"
    if ordinal:
        return f"<ord {_letters(int(value))} {lang}>"
    if isinstance(value, float):
        return f"<dec {_letters(repr(value))} {lang}>"
    return f"<card {_letters(int(value))} {lang}>"


class _Sent(str):
    pass


class FakeNLP:
    """What split_sentence needs of a spaCy pipeline: pipe_names, add_pipe, __call__ -> object with .sents."""
    sentences = None   # injected per call by the fixture generator

    def __init__(self):
        self.pipe_names = []

    def add_pipe(self, name):
        self.pipe_names.append(name)

    def __call__(self, text):
        sents = FakeNLP.sentences
        assert sents is not None, "inject the sentence list with set_sentences() first"
        return types.SimpleNamespace(sents=[_Sent(s) for s in sents])


def set_sentences(sentences):
    FakeNLP.sentences = list(sentences)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_loaded = None


def load_reference_tokenizer():
    """-> the reference's tokenizer module object (cached)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_text_available():
        raise FileNotFoundError(os.path.join(REF_ROOT, _TOK))
    saved = dict(sys.modules)
    try:
        _stub("pypinyin", pinyin=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("pypinyin stub")),
              Style=types.SimpleNamespace(TONE3=0))
        _stub("hangul_romanize", Transliter=lambda rule: types.SimpleNamespace(translit=lambda t: t))
        _stub("hangul_romanize.rule", academic=object())
        _stub("num2words", num2words=marker_num2words)
        _stub("cutlet", Cutlet=lambda: types.SimpleNamespace(romaji=lambda t: t))
        _stub("spacy")
        _stub("spacy.lang")
        for lang, cls in (("ar", "Arabic"), ("en", "English"), ("es", "Spanish"), ("ja", "Japanese"), ("zh", "Chinese")):
            _stub(f"spacy.lang.{lang}", **{cls: FakeNLP})
        # the vendored Chinese number normaliser is pure Python: load it from where it lies under its package path
        for pkg in ("auralis", "auralis.models", "auralis.models.xttsv2", "auralis.models.xttsv2.components",
                    "auralis.models.xttsv2.components.tts", "auralis.models.xttsv2.components.tts.layers",
                    "auralis.models.xttsv2.components.tts.layers.xtts"):
            if pkg not in sys.modules:
                m = types.ModuleType(pkg)
                m.__path__ = []
                sys.modules[pkg] = m
        zname = "auralis.models.xttsv2.components.tts.layers.xtts.zh_num2words"
        spec = importlib.util.spec_from_file_location(zname, os.path.join(REF_ROOT, _ZH))
        zmod = importlib.util.module_from_spec(spec)
        sys.modules[zname] = zmod
        spec.loader.exec_module(zmod)
        spec = importlib.util.spec_from_file_location("_ref_xtts_tokenizer", os.path.join(REF_ROOT, _TOK))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _loaded = mod
        return mod
    finally:
        for k in list(sys.modules):
            if k not in saved and (k == "auralis" or k.startswith(("spacy", "num2words", "pypinyin", "cutlet", "hangul_romanize", "auralis."))):
                del sys.modules[k]
