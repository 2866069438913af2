"""Generate tests/golden/facade_L30.npz: the oracle leg of the FACADE-level parity test (text in -> waveform out, 30 layers).

TEST INFRASTRUCTURE ONLY.  Usage (build container, ~20 s on 16 cores):  python -m oracle.make_golden_facade

What the drop-in promises is `TTS.generate_speech(TTSRequest(text=...))` (core/tts.py:310-355): language detection ->
`split_sentence` -> tokenizer -> one generation per <= 250-char chunk -> chunk outputs combined in order
(`TTSOutput.combine_outputs`).  The token-level goldens (c2 / c3 / ragged) submit ready-made ids to the engine; this fixture
walks the reference's per-request recipe (XTTSv2.py:690-814) on the CPU restatement instead:

  for every request:  language = get_language(text) when "auto" (requests.py __post_init__)
                      chunks   = tokenizer.batch_encode_with_split(text, language)      (tokenizer.py split_sentence + encode)
    for chunk i:      tokens   = GPTOracle.generate(cond + text ids; request sampling, seed + i, max_tokens = the checkpoint's
                                 gpt_max_audio_tokens, natural stop)                    (vllm_mm_gpt.py sample loop)
                      latents  = GPTOracle.second_pass_latents(...)                     (XTTSv2.py:617-687, literal second pass)
                      wav_i    = hifi_decoder_forward(latents, speaker embedding)       (hifigan_decoder.py:776-802)
                      output   = concat(wav_0, wav_1, ...)                              (TTSOutput.combine_outputs)

Requests: BASELINE configs[0]'s 50-char English sentence, greedy; three ~500-character paragraphs in English, French and German with
language="auto", sampled with the request defaults (T 0.75 / top_p 0.85 / top_k 50 / repetition penalty 5.0) and a fixed seed.
The checkpoint is the seeded synthetic one (seed 1234, 30 layers) with `mel_head.bias[1025]` raised so that the stop id is reachable
and `gpt_max_audio_tokens` = 96 in its config (keeps the CPU leg to minutes and the fixture small): some chunks end on the stop id,
the others at the cap.  Stored per request: detected language, chunk texts, chunk ids, chunk tokens, the combined waveform; the
test (tests/test_gpu_facade_parity.py) rebuilds the same checkpoint DIRECTORY with save_checkpoint and goes through TTS.from_pretrained.
The text side (split_sentence, cleaners, tokenizer) is the product's host code, which tests/test_text_golden.py pins to the
reference's own tokenizer.py functions; here it supplies the SAME chunking to both legs, and the test also asserts that the
facade chunked and encoded exactly as stored.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auralis_amd.api.requests import TTSRequest  # noqa: E402
from auralis_amd.api.text import XTTSTokenizer  # noqa: E402
from auralis_amd.checkpoint import make_synthetic_conditioning, make_synthetic_gpt, make_synthetic_xtts  # noqa: E402
from auralis_amd.config import XTTSDims  # noqa: E402
from oracle import xtts_oracle as O  # noqa: E402

N_LAYER = 30
STOP_BIAS = 1.0
STOP_BIAS_NATURAL = 0.8
MAX_TOKENS = 96
SEED = 77
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "facade_L30.npz")

C1_TEXT = "The quick brown fox jumps over the lazy dog today."
EN = ("It was a bright cold day in April, and the clocks were striking thirteen. Nobody in the street seemed to notice, and the wind "
      "kept pushing the dust along the old road as if nothing had happened at all. She counted twenty-three windows on the way to "
      "the station; Mr. Smith had said there would be more! Who could have known that the train would leave so early? The porter "
      "only shrugged, picked up the two heavy bags, and walked slowly towards the end of the platform without a word.")
FR = ("Il était une fois, dans une petite ville que nous ne connaissons pas, un homme qui avait beaucoup d'idées et très peu de temps "
      "pour les écrire. Il marchait chaque matin le long de la rivière avec son chien, et il parlait aux oiseaux comme à de vieux amis. "
      "Pourquoi ne restait-il jamais à la maison ? Personne ne le savait vraiment ; sa voisine disait seulement qu'il cherchait "
      "quelque chose qu'il avait perdu il y a longtemps, peut-être une lettre, peut-être un souvenir.")
DE = ("Es war einmal ein Mann, der nicht mit dem Zug fahren wollte und auch nicht zu Fuß gehen konnte, weil der Weg durch den Wald "
      "zu lang war. Also blieb er zu Hause und schrieb Briefe an seine Freunde. Jeden Abend zündete er eine Kerze an, setzte sich an "
      "den alten Tisch und dachte über die Jahre nach, die vergangen waren. Warum hatte er nie geantwortet, als sie ihn gefragt "
      "hatten? Vielleicht war es die Angst gewesen; vielleicht nur die Müdigkeit nach einem langen Winter.")

REQUESTS = [
    dict(name="c1", text=C1_TEXT, language="auto", temperature=0.0, seed=SEED),
    dict(name="en", text=EN, language="auto", seed=SEED + 100),
    dict(name="fr", text=FR, language="auto", seed=SEED + 200),
    dict(name="de", text=DE, language="auto", seed=SEED + 300),
]


def checkpoint(dims):
    """(gpt_sd, xtts_sd) of the fixture's checkpoint -- the test writes the same tensors with save_checkpoint(..., gpt_max_audio_tokens=MAX_TOKENS)."""
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=N_LAYER)
    gpt_sd["mel_head.bias"][1025] = STOP_BIAS
    xtts_sd = make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd)
    return gpt_sd, xtts_sd


def main_c1_natural():
    """tests/golden/facade_c1_natural_L30.npz: BASELINE configs[0] run the way a user runs it -- the 50-character sentence, greedy, the
    checkpoint's gpt_max_audio_tokens at the reference's own 605 (xttsv2_gpt_config.py: max_audio_tokens), natural stop: the sequence
    ends where the model emits the stop id (or at 605).  Usage: python -m oracle.make_golden_facade --c1-natural  (~1-2 min)."""
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    dims = XTTSDims()
    # stop bias 0.8 instead of the fixture's 1.0: the greedy sequence then runs 237 tokens before it emits the stop id (17 at 1.0) --
    # well past the 96-token cap of the other facade goldens, well short of 605: the stop itself ends it
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=N_LAYER)
    gpt_sd["mel_head.bias"][1025] = STOP_BIAS_NATURAL
    xtts_sd = make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd)
    cond, spk = make_synthetic_conditioning(dims)
    tok = XTTSTokenizer(None, vocab_size=xtts_sd["text_embedding.weight"].shape[0], synthetic=True)
    gpt = O.GPTOracle(gpt_sd, xtts_sd)
    w = O.vocoder_effective_weights(xtts_sd)
    req = TTSRequest(speaker_files=[], text=C1_TEXT, language="auto", temperature=0.0, seed=SEED)
    (ids,) = tok.batch_encode_with_split(req.text, req.language)
    c = gpt.build_cond(cond, ids)
    t0 = time.time()
    g = gpt.generate(c, O.SamplingCfg(temperature=0.0, top_k=req.top_k, top_p=req.top_p, repetition_penalty=req.repetition_penalty,
                                      max_tokens=605, ignore_stop=False, seed=req.seed & 0xFFFFFFFF))
    lat = gpt.second_pass_latents(c, g["tokens"])
    wav = O.hifi_decoder_forward(w, lat, spk).reshape(-1).numpy().astype(np.float32)
    print(f"c1 natural: {len(ids)} text ids -> {len(g['tokens'])} tokens{' (stop)' if g['tokens'][-1] == 1025 else ' (cap)'}, "
          f"{wav.shape[0]} samples, min margin {min(g['margins']) if 'margins' in g else float('nan'):.3e}, {time.time() - t0:.1f} s")
    out = os.path.join(os.path.dirname(OUT), "facade_c1_natural_L30.npz")
    np.savez_compressed(out, stop_bias=np.float32(STOP_BIAS_NATURAL), max_tokens=np.int32(605), n_layer=np.int32(N_LAYER), text=np.array(req.text),
                        language=np.array(req.language), seed=np.int64(req.seed), ids=np.asarray(ids, np.int32),
                        tokens=np.asarray(g["tokens"], np.int32), wav=wav)
    print(f"wrote {out} ({os.path.getsize(out) / 1e6:.2f} MB)")


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    dims = XTTSDims()
    gpt_sd, xtts_sd = checkpoint(dims)
    cond, spk = make_synthetic_conditioning(dims)
    tok = XTTSTokenizer(None, vocab_size=xtts_sd["text_embedding.weight"].shape[0], synthetic=True)
    gpt = O.GPTOracle(gpt_sd, xtts_sd)
    w = O.vocoder_effective_weights(xtts_sd)
    out = {"stop_bias": np.float32(STOP_BIAS), "max_tokens": np.int32(MAX_TOKENS), "n_layer": np.int32(N_LAYER),
           "names": np.array([r["name"] for r in REQUESTS])}
    t00 = time.time()
    for r in REQUESTS:
        kw = {k: v for k, v in r.items() if k != "name"}
        req = TTSRequest(speaker_files=[], **kw)      # __post_init__ detects the language exactly as the facade's request will
        from auralis_amd.api.text import split_sentence
        chunk_texts = split_sentence(req.text, req.language, tok.char_limit(req.language))
        chunk_ids = tok.batch_encode_with_split(req.text, req.language)
        assert len(chunk_texts) == len(chunk_ids)
        wavs, toks = [], []
        for i, ids in enumerate(chunk_ids):
            c = gpt.build_cond(cond, ids)
            cfg = O.SamplingCfg(temperature=req.temperature, top_k=req.top_k, top_p=req.top_p, repetition_penalty=req.repetition_penalty,
                                max_tokens=MAX_TOKENS, ignore_stop=False, seed=(req.seed + i) & 0xFFFFFFFF)
            t0 = time.time()
            g = gpt.generate(c, cfg)
            lat = gpt.second_pass_latents(c, g["tokens"])
            wav = O.hifi_decoder_forward(w, lat, spk).reshape(-1).numpy()
            toks.append(np.asarray(g["tokens"], np.int32))
            wavs.append(wav.astype(np.float32))
            print(f"{r['name']} [{req.language}] chunk {i}: {len(ids)} text ids -> {len(g['tokens'])} tokens"
                  f"{' (stop)' if g['tokens'][-1] == 1025 else ''}, {wav.shape[0]} samples, {time.time() - t0:.1f} s", flush=True)
        n = r["name"]
        out[f"{n}_text"] = np.array(req.text)
        out[f"{n}_language"] = np.array(req.language)
        out[f"{n}_seed"] = np.int64(req.seed)
        out[f"{n}_temperature"] = np.float32(req.temperature)
        out[f"{n}_n_chunks"] = np.int32(len(chunk_ids))
        out[f"{n}_chunk_texts"] = np.array(chunk_texts)
        for i in range(len(chunk_ids)):
            out[f"{n}_ids_{i}"] = np.asarray(chunk_ids[i], np.int32)
            out[f"{n}_tokens_{i}"] = toks[i]
        out[f"{n}_wav"] = np.concatenate(wavs)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 1e6:.2f} MB) in {time.time() - t00:.0f} s")


if __name__ == "__main__":
    if "--c1-natural" in sys.argv:
        main_c1_natural()
    else:
        main()
