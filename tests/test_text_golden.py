"""The text front-end (api/text.py, api/cleaners.py) against tests/golden/text_frontend.json: outputs of the reference's OWN
functions (src/auralis/models/xttsv2/config/tokenizer.py:51-236, 241-720, 805-820), produced by oracle/make_golden_text.py,
which runs them unmodified.  Bit-exact (strings / integers).

Not pinnable here, by name: spaCy's tokenizer + `sentencizer` (the sentence list of a split_sentence case is part of the
fixture; api/text.py's own _sentencize is only held to its stated rules in test_host_api.py), `num2words` (the fixture was
made with oracle.ref_text_import.marker_num2words standing in for it; this test passes the same marker to the product code,
so everything AROUND the spelling is pinned, the spelling itself is not), pypinyin / cutlet / hangul_romanize (zh / ja / ko
romanisation) and the vendored zh_num2words.TextNorm."""
import json
import os

import pytest

from auralis_amd.api import cleaners, text
from oracle.ref_text_import import marker_num2words, reference_text_available

GOLD = os.path.join(os.path.dirname(__file__), "golden", "text_frontend.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD, encoding="utf-8") as f:
        return json.load(f)


def test_find_best_split_point(gold):
    cases = gold["find_best_split_point"]
    assert len(cases) >= 200
    for c in cases:
        assert text.find_best_split_point(c["text"], c["target"], c["window"]) == c["pos"], c


def test_expand_abbreviations_every_language(gold):
    for c in gold["expand_abbreviations"]:
        assert cleaners.expand_abbreviations(c["text"], c["lang"]) == c["out"], c


def test_expand_symbols_every_language(gold):
    for c in gold["expand_symbols"]:
        assert cleaners.expand_symbols(c["text"], c["lang"]) == c["out"], c


def test_number_regex_plumbing_with_marker_speller(gold):
    """thousands separators, currency / decimal / ordinal / cardinal order, the integer-amount tail drop — per language"""
    cases = gold["expand_numbers_marker"]
    assert len(cases) >= 300
    for c in cases:
        assert cleaners.expand_numbers(c["text"], c["lang"], speller=marker_num2words) == c["out"], c


def test_multilingual_cleaners(gold):
    for c in gold["multilingual_cleaners_marker"]:
        assert cleaners.multilingual_cleaners(c["text"], c["lang"], speller=marker_num2words) == c["out"], c


def test_split_sentence_packing_loop_with_injected_sentences(gold):
    for c in gold["split_sentence_injected"]:
        assert text.split_sentence(c["text"], c["lang"], c["limit"], sentences=c["sentences"]) == c["out"], c


def test_preprocess_text(gold):
    for c in gold["preprocess_text"]:
        assert text.preprocess_text(c["text"], c["lang"]) == c["out"], c


@pytest.mark.skipif(not reference_text_available(), reason="needs /root/reference (build container only)")
def test_fixture_is_what_the_reference_produces_now(gold, tmp_path, monkeypatch):
    """regenerate the fixture from the reference and compare with the committed file"""
    import oracle.make_golden_text as mk
    out = tmp_path / "text_frontend.json"
    monkeypatch.setattr(mk, "OUT", str(out))
    mk.main()
    with open(out, encoding="utf-8") as f:
        assert json.load(f) == gold
