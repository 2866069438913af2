"""Language tag detection for TTSRequest(language="auto").

The reference calls langid.classify (src/auralis/common/definitions/requests.py:96-113), which is not available
offline.  This is a small script + stop-word detector covering the reference's supported tags; it is exact for
the en/fr/de mix of BASELINE config 5 on paragraph-sized inputs and falls back to "en".
"""
from __future__ import annotations

import re
from functools import lru_cache

SUPPORTED = ("en", "es", "fr", "de", "it", "pt", "pl", "tr", "ru", "nl", "cs", "ar", "zh-cn", "hu", "ko", "ja",
             "hi", "auto", "")

_STOP = {
    "en": "the and of to in is that it for with as was on are this be at by not have from or an they which you",
    "fr": "le la les de des et un une du en est que qui dans pour pas sur au ce il elle nous vous ne se avec son",
    "de": "der die das und ist nicht ein eine zu den mit von auf für dem sich des im auch als ich sie es wir wird",
    "es": "el la los las de y que en un una es por con no para su al lo como más pero sus le ya este sí porque",
    "it": "il lo la gli le di e che in un una è per non con del della sono più come ma anche si al dei nel",
    "pt": "o a os as de e que em um uma é para não com do da dos das por mais como mas ao se na no foi são",
    "nl": "de het een en van is dat niet op te in voor met zijn er aan ook als maar bij nog naar dan wat",
    "pl": "i w nie na z że się do jest to jak ale po co tak za od przez był dla czy tylko jego jej",
    "tr": "ve bir bu da de için ile ne çok daha gibi ama en kadar olarak var ben sen o mi mı",
    "cs": "a se na je že v to s z do o ale jako by pro tak po když jsem jsou byl být",
    "hu": "a az és hogy nem is egy meg de van ez volt már csak még mint én el ha",
}
_STOP = {k: frozenset(v.split()) for k, v in _STOP.items()}


@lru_cache(maxsize=1024)
def get_language(text: str) -> str:
    s = text[:2000]
    if re.search(r"[぀-ヿ]", s):
        return "ja"
    if re.search(r"[가-힯]", s):
        return "ko"
    if re.search(r"[一-鿿]", s):
        return "zh-cn"
    if re.search(r"[؀-ۿ]", s):
        return "ar"
    if re.search(r"[ऀ-ॿ]", s):
        return "hi"
    if re.search(r"[Ѐ-ӿ]", s):
        return "ru"
    words = re.findall(r"[^\W\d_]+", s.lower())
    if not words:
        return "en"
    best, score = "en", 0
    for lang, stop in _STOP.items():
        c = sum(1 for w in words if w in stop)
        if c > score:
            best, score = lang, c
    return best


def validate_language(language: str) -> str:
    if language not in SUPPORTED:
        raise ValueError(f"Language {language} not supported. Must be one of {SUPPORTED}")
    return language
