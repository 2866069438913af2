"""CPU: the facade-parity fixture (tests/golden/facade_L30.npz, oracle/make_golden_facade.py) against the product's host code --
the half of tests/test_gpu_facade_parity.py that needs no GPU: language detection, chunking and token ids of every stored request
are what api/requests.py / api/text.py produce today, and the stored outputs are self-consistent."""
import os

import numpy as np

from auralis_amd.api.requests import TTSRequest
from auralis_amd.api.text import XTTSTokenizer, split_sentence

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "facade_L30.npz")


def test_facade_fixture_matches_the_host_text_path(dims):
    g = np.load(GOLD)
    tok = XTTSTokenizer(None, vocab_size=dims.gpt.text_vocab, synthetic=True)
    assert g["names"].tolist() == ["c1", "en", "fr", "de"]
    for name in g["names"].tolist():
        req = TTSRequest(text=str(g[f"{name}_text"]), speaker_files=[], language="auto")
        assert req.language == str(g[f"{name}_language"])
        texts = split_sentence(req.text, req.language, tok.char_limit(req.language))
        assert texts == g[f"{name}_chunk_texts"].tolist()
        n = int(g[f"{name}_n_chunks"])
        ids = tok.batch_encode_with_split(req.text, req.language)
        assert [list(map(int, c)) for c in ids] == [g[f"{name}_ids_{i}"].tolist() for i in range(n)]
        total = 0
        for i in range(n):
            toks = g[f"{name}_tokens_{i}"]
            assert 1 <= len(toks) <= int(g["max_tokens"]) and ((toks >= 0) & (toks < 1026)).all()
            assert (toks[:-1] != 1025).all()                      # the stop id, when present, ends the chunk (XTTSv2.py:737)
            assert len(toks) == int(g["max_tokens"]) or toks[-1] == 1025
            total += dims.voc.samples_for_latents(len(toks))
        assert g[f"{name}_wav"].shape == (total,) and np.isfinite(g[f"{name}_wav"]).all()
    assert len(str(g["c1_text"])) == 50 and str(g["fr_language"]) == "fr" and str(g["de_language"]) == "de"
