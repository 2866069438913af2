"""GPU: a book-length text as ONE request through the facade (BASELINE configs[4]'s > 100 000-character path).

`TTS.generate_speech` cuts a text above 100 000 characters into several requests (`TTS.split_requests`, core/tts.py:236-255), each of
which `split_sentence` cuts into <= 250-character chunks; every chunk is one sequence on the engine, outputs come back in order.  Until
round 5 that path ran on the CPU fake engine only (tests/test_longform.py).  Here: 120 000 characters of English on a 2-layer real
engine with 64 slots (natural stop after a handful of tokens, so ~500 chunks take seconds):

  * the request is split in two, the chunks are the splitter's, the stream delivers them in order;
  * the non-streamed output is the concatenation of the streamed chunks, bit for bit (same seeds);
  * sampled chunks are equal, bit for bit, to the same ids synthesised ALONE with the chunk's seed: a chunk's audio does not depend on
    the other ~60 sequences it shared the engine with (batch invariance through the whole facade);
  * nothing is left behind: no tracked sequence, K/V blocks back, result blocks free.
"""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SENTENCES = [
    "It was a bright cold day in April, and the clocks were striking thirteen.",
    "Nobody in the street seemed to notice, and the wind kept pushing the dust along the old road as if nothing had happened at all.",
    "She counted twenty-three windows on the way to the station; the porter only shrugged and picked up the two heavy bags.",
    "Who could have known that the train would leave so early?",
    "The harbour lights came on one by one while the last ferry crossed the bay, and the gulls went quiet above the wall.",
]


def test_book_length_request_through_the_facade_on_the_real_engine(tmp_path, dims):
    from auralis_amd import TTS, TTSRequest
    from auralis_amd.checkpoint import make_synthetic_conditioning, make_synthetic_gpt, make_synthetic_xtts, save_checkpoint
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=2)
    gpt_sd["mel_head.bias"][1025] = 3.0
    save_checkpoint(str(tmp_path), gpt_sd, make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd), dims, synthetic_tokenizer=True)
    cond, spk = make_synthetic_conditioning(dims)
    voice = {"gpt_cond_latent": cond.numpy(), "speaker_embedding": spk.numpy()}
    rng = np.random.default_rng(11)
    parts, n = [], 0
    while n < 120_000:
        s = SENTENCES[int(rng.integers(len(SENTENCES)))]
        parts.append(s)
        n += len(s) + 1
    book = " ".join(parts)
    assert len(book) > 100_000
    tts = TTS(scheduler_max_concurrency=64).from_pretrained(str(tmp_path))
    try:
        eng = tts.tts_engine
        req = TTSRequest(text=book, speaker_files=[voice], language="en", seed=7)
        subs = TTS.split_requests(req)
        assert len(subs) == 2 and [len(s.text) for s in subs] == [100_000, len(book) - 100_000]
        ids = [c for s in subs for c in eng.tokenizer.batch_encode_with_split(s.text, "en")]
        seeds = [7 + i for s in subs for i in range(len(eng.tokenizer.batch_encode_with_split(s.text, "en")))]
        assert len(ids) > 400

        chunks = list(tts.generate_speech(TTSRequest(text=book, speaker_files=[voice], language="en", seed=7, stream=True)))
        assert len(chunks) == len(ids)
        assert all(len(c.array) > 0 and np.isfinite(c.array).all() for c in chunks)
        streamed = np.concatenate([c.array for c in chunks])
        whole = tts.generate_speech(req)
        assert whole.array.shape == streamed.shape and np.array_equal(whole.array, streamed)
        assert whole.token_length == sum(c.token_length for c in chunks)

        # a chunk alone == the chunk inside the book (same ids, same seed)
        orig = eng.tokenizer.batch_encode_with_split
        try:
            for k in (0, 1, len(ids) // 3, len(ids) // 2, len(ids) - 1):
                eng.tokenizer.batch_encode_with_split = lambda text, lang, k=k: [ids[k]]
                solo = tts.generate_speech(TTSRequest(text="x", speaker_files=[voice], language="en", seed=seeds[k]))
                assert solo.token_length == chunks[k].token_length, (k, solo.token_length, chunks[k].token_length)
                assert np.array_equal(solo.array, chunks[k].array), k
                del solo
        finally:
            eng.tokenizer.batch_encode_with_split = orig
        del chunks, whole, streamed
        gc.collect()
        st = eng.native.stats()
        assert st["sequences_tracked"] == 0 and st["kv_blocks_total"] - st["kv_blocks_free"] == 2
        assert st["result_blocks_free"] == st["result_blocks"] and eng.native._leased_bytes == 0
    finally:
        tts.close()
