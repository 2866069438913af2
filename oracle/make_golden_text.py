"""tests/golden/text_frontend.json: outputs of the reference's OWN text front-end functions (tokenizer.py, loaded unmodified
through oracle/ref_text_import.py) on seeded inputs.  Run in the build container: python -m oracle.make_golden_text

TEST INFRASTRUCTURE ONLY.  Sections: find_best_split_point (tokenizer.py:51-115), expand_abbreviations_multilingual /
expand_symbols_multilingual (:241-601), expand_numbers_multilingual with the marker num2words (:603-700),
multilingual_cleaners (:708-719), split_sentence's packing loop with an injected sentence list (:119-236),
XTTSTokenizerFast.preprocess_text (:805-820).  What the stubs leave unpinned is listed in ref_text_import.py."""
from __future__ import annotations

import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle.ref_text_import import load_reference_tokenizer, set_sentences  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "text_frontend.json")

_PIECES = [
    "The quick brown fox jumps over the lazy dog", "and then it rained for three days", "which nobody expected",
    "Le petit chat dort sur le canapé", "pendant que la pluie tombe", "Der alte Mann geht langsam über die Brücke",
    "weil es schon spät ist", "La casa es muy grande", "pero el jardín es pequeño", "Il treno parte alle otto",
    "a b c d e f g", "supercalifragilisticexpialidocious", "x", "هذا نص عربي قصير", "今天天气很好", "我们去公园吧",
    "これは日本語の文です", "Это русский текст", "mr. smith went to washington", "e.g. this one", "3.14159 is pi",
]
_SEPS = [". ", "! ", "? ", "؟ ", "။ ", "... ", ".\n\n", "\n\n", "\r\n\r\n", ": ", "; ", "；", "：", " | ", ", ", "，", "، ", "、",
         ") ", "] ", "】", "』 ", "» ", "》", " - ", " — ", "−", " & ", " + ", " = ", " / ", " ", "  ", "\t", "\n", ""]


def _rand_text(rng: random.Random, n_pieces: int) -> str:
    out = []
    for _ in range(n_pieces):
        out.append(rng.choice(_PIECES))
        out.append(rng.choice(_SEPS))
    return "".join(out)


def main() -> None:
    ref = load_reference_tokenizer()
    rng = random.Random(20260925)
    gold = {"source": "astramind-ai/Auralis src/auralis/models/xttsv2/config/tokenizer.py, run unmodified through "
                      "oracle/ref_text_import.py (spaCy / num2words / romanisers stubbed)"}

    # ---- A. find_best_split_point
    cases = []
    for i in range(260):
        text = _rand_text(rng, rng.randint(3, 14))
        window = rng.choice([30, 30, 30, 10, 50])
        target = rng.randint(0, max(1, len(text) - 1)) if i % 7 else rng.choice([0, len(text), len(text) + 5, window // 2])
        cases.append({"text": text, "target": target, "window": window,
                      "pos": ref.find_best_split_point(text, target, window)})
    gold["find_best_split_point"] = cases

    # ---- B / C. abbreviations and symbols per language
    langs = sorted(ref._symbols_multilingual.keys())
    abbr, sym = [], []
    for lang in langs:
        words = []
        for rx, _ in ref._abbreviations.get(lang, []):
            w = rx.pattern.replace("\\b", "").replace("\\.", ".")
            words += [w, w.upper(), w.capitalize(), "x" + w, w.rstrip(".") + " "]
        text = " | ".join(words) + " | plain words dr mr. st. co. ltd. end"
        abbr.append({"lang": lang, "text": text, "out": ref.expand_abbreviations_multilingual(text, lang)})
        text = "a&b @home 50% #1 $5 £7 20° & && a  b   c  "
        sym.append({"lang": lang, "text": text, "out": ref.expand_symbols_multilingual(text, lang)})
    abbr.append({"lang": "xx", "text": "dr. who", "out": ref.expand_abbreviations_multilingual("dr. who", "xx")})
    sym.append({"lang": "xx", "text": " a & b ", "out": ref.expand_symbols_multilingual(" a & b ", "xx")})
    gold["expand_abbreviations"] = abbr
    gold["expand_symbols"] = sym

    # ---- D. numbers (marker num2words): regex plumbing per language
    num_texts = [
        "1,234 and 1.234 and 12,345,678.9 and 12.345.678,9", "pi is 3.14 or 3,14", "it costs $12 or $12.50 or 12$ or $1,000",
        "prix 12€ ou 12,50€ ou €7 ou 1.000€", "£3 and £3.05 and 3£", "the 1st 2nd 3rd 4th 21st", "le 1er 2e 3ème 4º 5ª", "am 3. mai und 4. ",
        "el 1o la 2a los 3os", "10-й 2-го", "3de 4ste 5e", "5inci 6. 7nci", "8. 9adik 10ik", "1번째 2번", "0 7 42 100 1000000",
        "v2 x86 a1b2", "no digits here", "3.5% of $2", "12:30", "1,5 2.5,3", "$ 5 and 5 $", "",
    ]
    nums = []
    for lang in [l for l in langs if l != "zh"] + ["xx"]:
        for t in num_texts:
            nums.append({"lang": lang, "text": t, "out": ref.expand_numbers_multilingual(t, lang)})
    gold["expand_numbers_marker"] = nums

    # ---- E. multilingual_cleaners
    clean_texts = {
        "en": ['Dr. Smith & Mrs. Jones said "hello" to the Rev. Brown @ St. Mary\'s;   100% sure, #1!', "MR. X paid $12.50 for 1,234 items on the 2nd."],
        "fr": ['Mme. Dupont et le Dr. Martin ont dit "bonjour" à St. Denis & Co.', "Elle a payé 12,50€ pour 1.234 articles le 1er mai."],
        "de": ['Fr. Müller und Dr. Schmidt sagten "Hallo" bei der Co. in St. Gallen.', "Er zahlte 12,50€ für 1.234 Artikel am 3. Mai."],
        "es": ["La Sra. García y el Dr. López   dijeron hola & adiós.", "Pagó 12,50€ el 1o de mayo."],
        "it": ["Il Sig. Rossi e il Dr. Bianchi @ casa.", "Ha pagato 12,50€."], "pt": ["A Sra. Silva e o Dr. Costa # um.", "Pagou 12,50€."],
        "pl": ["P. Kowalska i Dr. Nowak 50% razy.", "Zapłacił 12,50€."], "tr": ["İstanbul'da B. Yılmaz ve Dr. Öz ÜNLÜ.", "12,50€ 3.5 5inci"],
        "ru": ["Г-жа Иванова и д-р Петров & Ко.", "1,234 и 2-го"], "nl": ["Dhr. Jansen en Mevr. De Vries @ huis.", "3de keer 12,50€"],
        "cs": ["Dr. Novák a Ing. Svoboda & spol.", "3. května 12,50€"], "ar": ["هذا نص & اختبار 50%", "12 و 3.5"],
        "hu": ["Dr. Nagy és B. Kovács # egy.", "3. nap 12,50€"], "ko": ["안녕하세요 & 감사합니다 50%", "1번째 12"],
    }
    cl = []
    for lang, ts in clean_texts.items():
        for t in ts:
            cl.append({"lang": lang, "text": t, "out": ref.multilingual_cleaners(t, lang)})
    gold["multilingual_cleaners_marker"] = cl

    # ---- F. split_sentence packing loop (sentence list injected: it is part of the case)
    from auralis_amd.api.text import _sentencize
    sp = []
    for i in range(60):
        lang = rng.choice(["en", "en", "fr", "de", "es", "zh", "ar"])
        text = _rand_text(rng, rng.randint(6, 40)).strip()
        limit = rng.choice([250, 273, 253, 82, 60, 40, 120])
        sents = _sentencize(text, lang) if i % 5 else [text]   # every 5th: one giant "sentence"
        set_sentences(sents)
        sp.append({"lang": lang, "text": text, "limit": limit, "sentences": sents, "out": ref.split_sentence(text, lang, limit)})
    gold["split_sentence_injected"] = sp

    # ---- G. preprocess_text (languages without romanisation)
    pp = []
    for lang, t in (("en", 'Dr.  Smith   SAID "hi" & left'), ("fr-fr", "Mme. Dupont  & Co."), ("xx", "  Mixed   CASE\ttext 5 "), ("de", "Fr. Müller @ Haus")):
        pp.append({"lang": lang, "text": t, "out": ref.XTTSTokenizerFast.preprocess_text(None, t, lang)})
    gold["preprocess_text"] = pp

    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(gold, f, ensure_ascii=False, indent=0)
    print(f"wrote {OUT}: " + ", ".join(f"{k}={len(v)}" for k, v in gold.items() if isinstance(v, list)))


if __name__ == "__main__":
    main()
