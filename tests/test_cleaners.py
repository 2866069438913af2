"""CPU: text normalisation (api/cleaners.py) — pipeline order and num2words-style spelling for en / fr / de.
Expected strings follow num2words' documented conventions (not generated here: num2words is absent offline)."""
import pytest

from auralis_amd.api import cleaners as Cl


@pytest.mark.parametrize("n,w", [(0, "zero"), (7, "seven"), (13, "thirteen"), (21, "twenty-one"), (100, "one hundred"),
                                 (101, "one hundred and one"), (999, "nine hundred and ninety-nine"), (1000, "one thousand"),
                                 (1001, "one thousand and one"), (1234, "one thousand, two hundred and thirty-four"),
                                 (1000000, "one million"), (1200000, "one million, two hundred thousand")])
def test_english_cardinals(n, w):
    assert Cl._en_card(n) == w


@pytest.mark.parametrize("n,w", [(1, "first"), (2, "second"), (3, "third"), (5, "fifth"), (12, "twelfth"), (20, "twentieth"),
                                 (21, "twenty-first"), (100, "one hundredth"), (101, "one hundred and first")])
def test_english_ordinals(n, w):
    assert Cl._en_ord(n) == w


@pytest.mark.parametrize("n,w", [(16, "seize"), (21, "vingt et un"), (22, "vingt-deux"), (70, "soixante-dix"), (71, "soixante et onze"),
                                 (80, "quatre-vingts"), (81, "quatre-vingt-un"), (99, "quatre-vingt-dix-neuf"), (100, "cent"),
                                 (200, "deux cents"), (201, "deux cent un"), (1000, "mille"), (2000, "deux mille"),
                                 (1999, "mille neuf cent quatre-vingt-dix-neuf"), (1000000, "un million")])
def test_french_cardinals(n, w):
    assert Cl._fr_card(n) == w


@pytest.mark.parametrize("n,w", [(1, "eins"), (11, "elf"), (16, "sechzehn"), (21, "einundzwanzig"), (30, "dreißig"), (100, "einhundert"),
                                 (101, "einhunderteins"), (1000, "eintausend"), (1984, "eintausendneunhundertvierundachtzig"),
                                 (1000000, "eine million")])
def test_german_cardinals(n, w):
    assert Cl._de_card(n) == w


def test_ordinals_fr_de():
    assert [Cl._fr_ord(n) for n in (1, 2, 4, 5, 9, 21)] == ["premier", "deuxième", "quatrième", "cinquième", "neuvième", "vingt et unième"]
    assert [Cl._de_ord(n) for n in (1, 2, 3, 7, 19, 20, 21)] == ["erste", "zweite", "dritte", "siebte", "neunzehnte", "zwanzigste", "einundzwanzigste"]


def test_pipeline_english():
    out = Cl.multilingual_cleaners('Dr. Smith paid $12.50 for 3 "apples" & 1,250 pears on the 2nd;  that is 50% more!', "en")
    assert out == ("doctor smith paid twelve dollars, fifty cents for three apples and one thousand, two hundred and fifty pears "
                   "on the second; that is fifty percent more!")
    assert Cl.multilingual_cleaners("It costs 3.14 or £5", "en") == "it costs three point one four or five pounds"


def test_pipeline_french_german():
    assert Cl.multilingual_cleaners("Mme. Dupont a 21 ans et 1.500 livres", "fr") == "madame dupont a vingt et un ans et mille cinq cents livres"
    assert Cl.multilingual_cleaners("Dr. Weiß hat 12,50€ & 3 Hunde", "de") == "doktor weiß hat zwölf euro und fünfzig cent und drei hunde"


def test_other_languages_keep_digits_and_collapse_whitespace():
    assert Cl.multilingual_cleaners("Hola   mundo 42 %", "es") == "hola mundo 42 por ciento"
