"""Text normalisation in front of the BPE tokenizer: quotes -> lowercase -> numbers -> abbreviations -> symbols ->
whitespace, the order of the reference's `multilingual_cleaners` (src/auralis/models/xttsv2/config/tokenizer.py:
708-719; number handling 603-700, abbreviation / symbol tables 241-600).

Everything except the spelling of a number is pinned to the reference's own functions by tests/golden/text_frontend.json
(oracle/make_golden_text.py runs them unmodified): the tables, the regexes and their order, the separator removal, the
integer-amount tail drop of currencies.  The reference delegates the spelling itself to `num2words`; when that package is
importable it is used exactly as the reference uses it, otherwise en / fr / de — the languages of BASELINE config 5 — are
spelled out here following num2words' conventions ("one thousand, two hundred and thirty-four", "quatre-vingt-un",
"einundzwanzig") and other languages keep their digits.  zh numbers (the reference's vendored zh_num2words.TextNorm) are
not restated.  CPU text plumbing in front of the hot path.
"""
from __future__ import annotations

import re
from typing import Callable, Dict, List, Tuple

_WS = re.compile(r"\s+")

# ------------------------------------------------------------------------------------------------ tables
_ABBREV = {   # tokenizer.py:241-399
    "en": [("mrs", "misess"), ("mr", "mister"), ("dr", "doctor"), ("st", "saint"), ("co", "company"), ("jr", "junior"),
           ("maj", "major"), ("gen", "general"), ("drs", "doctors"), ("rev", "reverend"), ("lt", "lieutenant"),
           ("hon", "honorable"), ("sgt", "sergeant"), ("capt", "captain"), ("esq", "esquire"), ("ltd", "limited"),
           ("col", "colonel"), ("ft", "fort")],
    "es": [("sra", "señora"), ("sr", "señor"), ("dr", "doctor"), ("dra", "doctora"), ("st", "santo"), ("co", "compañía"),
           ("jr", "junior"), ("ltd", "limitada")],
    "fr": [("mme", "madame"), ("mr", "monsieur"), ("dr", "docteur"), ("st", "saint"), ("co", "compagnie"), ("jr", "junior"),
           ("ltd", "limitée")],
    "de": [("fr", "frau"), ("dr", "doktor"), ("st", "sankt"), ("co", "firma"), ("jr", "junior")],
    "pt": [("sra", "senhora"), ("sr", "senhor"), ("dr", "doutor"), ("dra", "doutora"), ("st", "santo"), ("co", "companhia"),
           ("jr", "júnior"), ("ltd", "limitada")],
    "it": [("sig", "signore"), ("dr", "dottore"), ("st", "santo"), ("co", "compagnia"), ("jr", "junior"), ("ltd", "limitata")],
    "pl": [("p", "pani"), ("m", "pan"), ("dr", "doktor"), ("sw", "święty"), ("jr", "junior")],
    "cs": [("dr", "doktor"), ("ing", "inženýr"), ("p", "pan")],
    "nl": [("dhr", "de heer"), ("mevr", "mevrouw"), ("dr", "dokter"), ("jhr", "jonkheer")],
    "tr": [("b", "bay"), ("byk", "büyük"), ("dr", "doktor")],
    "hu": [("dr", "doktor"), ("b", "bácsi"), ("nőv", "nővér")],
}
_ABBREV_RE = {lang: [(re.compile(r"\b%s\." % a, re.IGNORECASE), b) for a, b in lst] for lang, lst in _ABBREV.items()}
# The same substitutions in ONE pass per text: the abbreviations of a language are distinct words, each pattern is `\b word \.`, and
# no replacement contains a period, so no substitution can create, destroy or overlap another one's match -- applying them one
# after the other (the reference's loop) and all at once give the same string.  (18 regex passes per chunk were a quarter of the
# facade's per-request time.)  A match the table cannot resolve (case folding of exotic letters) falls back to the loop.
_ABBREV_ONE = {lang: (re.compile(r"\b(%s)\." % "|".join(sorted((a for a, _ in lst), key=len, reverse=True)), re.IGNORECASE), dict(lst))
               for lang, lst in _ABBREV.items()}
# Russian abbreviations carry a hyphen and no period (tokenizer.py:365-372)
_ABBREV_RE["ru"] = [(re.compile(r"\b%s\b" % a, re.IGNORECASE), b) for a, b in (("г-жа", "госпожа"), ("г-н", "господин"), ("д-р", "доктор"))]

_SYMBOL_WORDS = {   # & @ % # $ £ °  (tokenizer.py:407-594)
    "en": ("and", "at", "percent", "hash", "dollar", "pound", "degree"),
    "es": ("y", "arroba", "por ciento", "numeral", "dolar", "libra", "grados"),
    "fr": ("et", "arobase", "pour cent", "dièse", "dollar", "livre", "degrés"),
    "de": ("und", "at", "prozent", "raute", "dollar", "pfund", "grad"),
    "pt": ("e", "arroba", "por cento", "cardinal", "dólar", "libra", "graus"),
    "it": ("e", "chiocciola", "per cento", "cancelletto", "dollaro", "sterlina", "gradi"),
    "pl": ("i", "małpa", "procent", "krzyżyk", "dolar", "funt", "stopnie"),
    "ar": ("و", "على", "في المئة", "رقم", "دولار", "جنيه", "درجة"),
    "zh": ("和", "在", "百分之", "号", "美元", "英镑", "度"),
    "cs": ("a", "na", "procento", "křížek", "dolar", "libra", "stupně"),
    "ru": ("и", "собака", "процентов", "номер", "доллар", "фунт", "градус"),
    "nl": ("en", "bij", "procent", "hekje", "dollar", "pond", "graden"),
    "tr": ("ve", "at", "yüzde", "diyez", "dolar", "sterlin", "derece"),
    "hu": ("és", "kukac", "százalék", "kettőskereszt", "dollár", "font", "fok"),
    "ko": ("그리고", "에", "퍼센트", "번호", "달러", "파운드", "도"),
}
_SYMBOLS = {lang: [(sym, " %s " % w) for sym, w in zip("&@%#$£°", words)] for lang, words in _SYMBOL_WORDS.items()}

_ORDINAL_RE = {   # tokenizer.py:603-618
    "en": re.compile(r"([0-9]+)(st|nd|rd|th)"),
    "es": re.compile(r"([0-9]+)(º|ª|er|o|a|os|as)"),
    "fr": re.compile(r"([0-9]+)(º|ª|er|re|e|ème)"),
    "de": re.compile(r"([0-9]+)(st|nd|rd|th|º|ª|\.(?=\s|$))"),
    "pt": re.compile(r"([0-9]+)(º|ª|o|a|os|as)"),
    "it": re.compile(r"([0-9]+)(º|°|ª|o|a|i|e)"),
    "pl": re.compile(r"([0-9]+)(º|ª|st|nd|rd|th)"),
    "ar": re.compile(r"([0-9]+)(ون|ين|ث|ر|ى)"),
    "cs": re.compile(r"([0-9]+)\.(?=\s|$)"),
    "ru": re.compile(r"([0-9]+)(-й|-я|-е|-ое|-ье|-го)"),
    "nl": re.compile(r"([0-9]+)(de|ste|e)"),
    "tr": re.compile(r"([0-9]+)(\.|inci|nci|uncu|üncü|\.)"),
    "hu": re.compile(r"([0-9]+)(\.|adik|edik|odik|edik|ödik|ödike|ik)"),
    "ko": re.compile(r"([0-9]+)(번째|번|차|째)"),
}
_NUMBER_RE = re.compile(r"[0-9]+")
_CURRENCY_RE = {
    "GBP": re.compile(r"((£[0-9\.\,]*[0-9]+)|([0-9\.\,]*[0-9]+£))"),
    "USD": re.compile(r"((\$[0-9\.\,]*[0-9]+)|([0-9\.\,]*[0-9]+\$))"),
    "EUR": re.compile(r"(([0-9\.\,]*[0-9]+€)|((€[0-9\.\,]*[0-9]+)))"),
}
_COMMA_NUMBER_RE = re.compile(r"\b\d{1,3}(,\d{3})*(\.\d+)?\b")
_DOT_NUMBER_RE = re.compile(r"\b\d{1,3}(\.\d{3})*(\,\d+)?\b")
_DECIMAL_RE = re.compile(r"([0-9]+[.,][0-9]+)")
_ANY_DIGIT = re.compile(r"\d")


# ------------------------------------------------------------------------------------------------ number spelling
def _en_card(n: int) -> str:
    ones = ["zero", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven", "twelve",
            "thirteen", "fourteen", "fifteen", "sixteen", "seventeen", "eighteen", "nineteen"]
    tens = ["", "", "twenty", "thirty", "forty", "fifty", "sixty", "seventy", "eighty", "ninety"]
    if n < 20:
        return ones[n]
    if n < 100:
        t, o = divmod(n, 10)
        return tens[t] + ("-" + ones[o] if o else "")
    if n < 1000:
        h, r = divmod(n, 100)
        return ones[h] + " hundred" + (" and " + _en_card(r) if r else "")
    parts: List[str] = []
    for value, name in ((10 ** 12, "trillion"), (10 ** 9, "billion"), (10 ** 6, "million"), (1000, "thousand")):
        if n >= value:
            q, n = divmod(n, value)
            parts.append(_en_card(q) + " " + name)
    if n:
        parts.append(("and " if n < 100 else "") + _en_card(n))
    out = parts[0]
    for p in parts[1:]:
        out += (" " if p.startswith("and ") else ", ") + p
    return out


def _en_ord(n: int) -> str:
    irregular = {"one": "first", "two": "second", "three": "third", "five": "fifth", "eight": "eighth", "nine": "ninth",
                 "twelve": "twelfth"}
    words = _en_card(n)
    head, sep, last = words.rpartition("-") if "-" in words.split(" ")[-1] else words.rpartition(" ")
    if last in irregular:
        last = irregular[last]
    elif last.endswith("y"):
        last = last[:-1] + "ieth"
    else:
        last = last + "th"
    return head + sep + last


def _fr_card(n: int) -> str:
    ones = ["zéro", "un", "deux", "trois", "quatre", "cinq", "six", "sept", "huit", "neuf", "dix", "onze", "douze", "treize",
            "quatorze", "quinze", "seize", "dix-sept", "dix-huit", "dix-neuf"]
    tens = {20: "vingt", 30: "trente", 40: "quarante", 50: "cinquante", 60: "soixante"}
    if n < 20:
        return ones[n]
    if n < 70:
        t, o = divmod(n, 10)
        if o == 0:
            return tens[t * 10]
        return tens[t * 10] + (" et un" if o == 1 else "-" + ones[o])
    if n < 80:
        return "soixante" + (" et onze" if n == 71 else "-" + ones[n - 60])
    if n < 100:
        return "quatre-vingts" if n == 80 else "quatre-vingt-" + ones[n - 80]
    if n < 1000:
        h, r = divmod(n, 100)
        head = "cent" if h == 1 else ones[h] + " cent" + ("s" if r == 0 else "")
        return head + (" " + _fr_card(r) if r else "")
    if n < 10 ** 6:
        q, r = divmod(n, 1000)
        head = "mille" if q == 1 else _fr_card(q).replace("quatre-vingts", "quatre-vingt").replace(" cents", " cent") + " mille"
        return head + (" " + _fr_card(r) if r else "")
    for value, name in ((10 ** 9, "milliard"), (10 ** 6, "million")):
        if n >= value:
            q, r = divmod(n, value)
            return _fr_card(q) + " " + name + ("s" if q > 1 else "") + (" " + _fr_card(r) if r else "")
    return str(n)


def _fr_ord(n: int) -> str:
    if n == 1:
        return "premier"
    w = _fr_card(n)
    if w.endswith("e"):
        w = w[:-1]
    elif w.endswith("f"):
        w = w[:-1] + "v"
    elif w.endswith("q"):
        w = w + "u"
    elif w.endswith("s") and (w.endswith("cents") or w.endswith("vingts")):
        w = w[:-1]
    return w + "ième"


def _de_card(n: int, standalone: bool = True) -> str:
    ones = ["null", "ein", "zwei", "drei", "vier", "fünf", "sechs", "sieben", "acht", "neun", "zehn", "elf", "zwölf", "dreizehn",
            "vierzehn", "fünfzehn", "sechzehn", "siebzehn", "achtzehn", "neunzehn"]
    tens = ["", "", "zwanzig", "dreißig", "vierzig", "fünfzig", "sechzig", "siebzig", "achtzig", "neunzig"]
    if n == 1:
        return "eins" if standalone else "ein"
    if n < 20:
        return ones[n]
    if n < 100:
        t, o = divmod(n, 10)
        return (ones[o] + "und" if o else "") + tens[t]
    if n < 1000:
        h, r = divmod(n, 100)
        return ones[h] + "hundert" + (_de_card(r, standalone) if r else "")
    if n < 10 ** 6:
        q, r = divmod(n, 1000)
        return _de_card(q, False) + "tausend" + (_de_card(r, standalone) if r else "")
    for value, sing, plur in ((10 ** 9, "milliarde", "milliarden"), (10 ** 6, "million", "millionen")):
        if n >= value:
            q, r = divmod(n, value)
            head = "eine " + sing if q == 1 else _de_card(q, False) + " " + plur
            return head + (" " + _de_card(r, standalone) if r else "")
    return str(n)


def _de_ord(n: int) -> str:
    special = {1: "erste", 3: "dritte", 7: "siebte", 8: "achte"}
    if n in special:
        return special[n]
    if n < 20:
        return _de_card(n) + "te"
    if n < 100 or n % 100 == 0 or n % 100 >= 20:
        tail = n % 100
        if 0 < tail < 20 and n >= 100:
            return _de_card(n - tail, False) + _de_ord(tail)
        return _de_card(n, False) + "ste"
    return _de_card(n - n % 100, False) + _de_ord(n % 100)


_CARD: Dict[str, Callable[[int], str]] = {"en": _en_card, "fr": _fr_card, "de": _de_card}
_ORD: Dict[str, Callable[[int], str]] = {"en": _en_ord, "fr": _fr_ord, "de": _de_ord}
_POINT = {"en": "point", "fr": "virgule", "de": "komma"}
_CURRENCY_WORDS = {   # (major singular, major plural, minor singular, minor plural)
    "en": {"USD": ("dollar", "dollars", "cent", "cents"), "GBP": ("pound", "pounds", "penny", "pence"),
           "EUR": ("euro", "euro", "cent", "cents")},
    "fr": {"USD": ("dollar", "dollars", "cent", "cents"), "GBP": ("livre", "livres", "penny", "pence"),
           "EUR": ("euro", "euros", "centime", "centimes")},
    "de": {"USD": ("dollar", "dollar", "cent", "cent"), "GBP": ("pfund", "pfund", "penny", "pence"),
           "EUR": ("euro", "euro", "cent", "cent")},
}
# joiner between the major and the minor unit of num2words' currency strings (tokenizer.py:651-666); the reference cuts an
# integer amount at the LAST occurrence of it ("twelve dollars, zero cents" -> "twelve dollars")
_AND_EQUIVALENTS = {"en": ", ", "es": " con ", "fr": " et ", "de": " und ", "pt": " e ", "it": " e ", "pl": ", ", "cs": ", ",
                    "ru": ", ", "nl": ", ", "ar": ", ", "tr": ", ", "hu": ", ", "ko": ", "}

# A speller has num2words' call signature: speller(value, lang=..., to="cardinal" | "currency", ordinal=False, currency=None).
Speller = Callable[..., str]


def builtin_speller(value, lang="en", to="cardinal", ordinal=False, currency=None) -> str:
    """en / fr / de spelling in num2words' conventions (see the module docstring); raises NotImplementedError otherwise."""
    if lang not in _CARD:
        raise NotImplementedError(lang)
    if to == "currency":
        major = int(value)
        cents = int(round((value - major) * 100))
        ms, mp, cs, cp = _CURRENCY_WORDS[lang][currency]
        return (f"{_CARD[lang](major)} {ms if major == 1 else mp}{_AND_EQUIVALENTS[lang]}"
                f"{_CARD[lang](cents)} {cs if cents == 1 else cp}")
    if ordinal:
        return _ORD[lang](int(value))
    if isinstance(value, float):
        whole, frac = repr(value).split(".")
        return f"{_CARD[lang](int(whole))} {_POINT[lang]} " + " ".join(_CARD[lang](int(d)) for d in frac)
    return _CARD[lang](int(value))


_NUM2WORDS = []   # [callable or None], filled on first use: a failed import costs ~0.3 ms of path search EVERY time it is retried


def default_speller(lang: str):
    """num2words itself when it is installed (what the reference calls), else the built-in en / fr / de speller, else None
    (digits stay)."""
    if not _NUM2WORDS:
        try:
            from num2words import num2words
            _NUM2WORDS.append(num2words)
        except ImportError:
            _NUM2WORDS.append(None)
    if _NUM2WORDS[0] is not None:
        return _NUM2WORDS[0]
    return builtin_speller if lang in _CARD else None


def expand_numbers(text: str, lang: str, speller: "Speller | None" = None) -> str:
    """expand_numbers_multilingual (tokenizer.py:681-700) for the non-zh languages: thousands separators removed ("," for
    en / ru, "." otherwise), then currencies (GBP, USD, EUR; failures ignored as in the reference), decimals (not for tr),
    ordinals (languages with a pattern), remaining integers."""
    base = lang.split("-")[0]
    if base == "zh":
        return text   # the reference's vendored Chinese normaliser (zh_num2words.TextNorm) is not restated
    if _ANY_DIGIT.search(text) is None:
        return text   # every pattern below needs a digit: nothing to do (and nine regex passes less per chunk)
    spell = speller if speller is not None else default_speller(base)
    if spell is None:
        return text
    n2w_lang = "cz" if base == "cs" else base   # tokenizer.py:645: num2words calls Czech "cz"

    def currency(m: "re.Match", cur: str) -> str:
        amount = float(re.sub(r"[^\d.]", "", m.group(0).replace(",", ".")))
        full = spell(amount, to="currency", currency=cur, lang=n2w_lang)
        if amount.is_integer():
            last = full.rfind(_AND_EQUIVALENTS.get(base, ", "))
            if last != -1:
                full = full[:last]
        return full

    if base in ("en", "ru"):
        text = _COMMA_NUMBER_RE.sub(lambda m: m.group(0).replace(",", ""), text)
    else:
        text = _DOT_NUMBER_RE.sub(lambda m: m.group(0).replace(".", ""), text)
    try:
        for cur in ("GBP", "USD", "EUR"):
            text = _CURRENCY_RE[cur].sub(lambda m, c=cur: currency(m, c), text)
    except Exception:
        pass
    if base != "tr":
        text = _DECIMAL_RE.sub(lambda m: spell(float(m.group(1).replace(",", ".")), lang=n2w_lang), text)
    if base in _ORDINAL_RE:
        text = _ORDINAL_RE[base].sub(lambda m: spell(int(m.group(1)), ordinal=True, lang=n2w_lang), text)
    return _NUMBER_RE.sub(lambda m: spell(int(m.group(0)), lang=n2w_lang), text)


def expand_abbreviations(text: str, lang: str) -> str:
    base = lang.split("-")[0]
    one = _ABBREV_ONE.get(base)
    if one is not None:
        if "." not in text:
            return text
        rx, table = one
        try:
            return rx.sub(lambda m: table[m.group(1).lower()], text)
        except KeyError:
            pass
    for rx, rep in _ABBREV_RE.get(base, []):
        text = rx.sub(rep, text)
    return text


def expand_symbols(text: str, lang: str) -> str:
    for sym, rep in _SYMBOLS.get(lang.split("-")[0], []):
        text = text.replace(sym, rep).replace("  ", " ")
    return text.strip()


def multilingual_cleaners(text: str, lang: str, speller: "Speller | None" = None) -> str:
    text = text.replace('"', "")
    if lang == "tr":
        text = text.replace("İ", "i").replace("Ö", "ö").replace("Ü", "ü")
    text = text.lower()
    text = expand_numbers(text, lang, speller)
    text = expand_abbreviations(text, lang)
    text = expand_symbols(text, lang)
    return _WS.sub(" ", text)
