"""Where each BASELINE.json config is exercised.

  configs[0]  50-char English utterance, greedy, reference CPU path        -> test_config0_cpu_reference_path (here, CPU)
  configs[1]  200-char utterance, greedy, 1 x MI355X, batch 1              -> tests/test_gpu_gpt.py (ids bit-exact, 3 and 30 layers),
                                                                             tests/test_gpu_vocoder.py, tests/test_gpu_edges.py
  configs[2]  64 concurrent utterances, T 0.75 / top_p 0.85, one GPU       -> bench.py (the metric), test_gpu_gpt.py batch invariance +
                                                                             sampled-token parity, test_gpu_edges.py pipelined decode
  configs[3]  512 utterances, 64 per GPU, RCCL-broadcast speaker latent    -> tests/test_parallel_cpu.py (world_size 2, gloo); bench.py --gpus N
  configs[4]  book-length stream, mixed en/fr/de, chunk-ordered            -> tests/test_longform.py (CPU fake engine, gloo, GPU at test scale)
"""
import numpy as np
import torch

from auralis_amd.api.lang import get_language
from auralis_amd.api.text import XTTSTokenizer, split_sentence
from auralis_amd.checkpoint import make_synthetic_conditioning, make_synthetic_gpt, make_synthetic_xtts
from oracle import xtts_oracle as O


def test_config0_cpu_reference_path(dims):
    """The whole path on the CPU restatement of the reference (2 GPT layers to keep it fast): text -> ids -> greedy mel
    tokens -> second-pass latents (XTTSv2.py:617-687) -> HiFi-GAN -> 24 kHz waveform.  Deterministic, finite, right length."""
    text = "The quick brown fox jumps over the lazy dog today."
    assert len(text) == 50 and get_language(text) == "en"
    tok = XTTSTokenizer(None, synthetic=True)
    chunks = split_sentence(text, "en", tok.char_limit("en"))
    assert chunks == [text]
    ids = tok.encode_chunk(chunks[0], "en")
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=2)
    xtts_sd = make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd)
    cond, spk = make_synthetic_conditioning(dims)
    gpt = O.GPTOracle(gpt_sd, xtts_sd)
    c = gpt.build_cond(cond, ids)
    cfg = O.SamplingCfg(temperature=0.0, max_tokens=6, ignore_stop=True)
    out = gpt.generate(c, cfg)
    again = gpt.generate(c, cfg)
    assert out["tokens"] == again["tokens"] and len(out["tokens"]) == 6 and all(0 <= t < 1026 for t in out["tokens"])
    lat = gpt.second_pass_latents(c, out["tokens"])
    assert lat.shape == (1, 6, 1024)
    wav = O.hifi_decoder_forward(O.vocoder_effective_weights(xtts_sd), lat, spk).reshape(-1)
    assert wav.shape[0] == dims.voc.samples_for_latents(6) and torch.isfinite(wav).all() and float(wav.abs().max()) <= 1.0
    assert np.isclose(wav.shape[0] / 24000.0, 6 * 1024 / 22050.0, rtol=0.05)     # 1024-sample hop at 22.05 kHz per mel token
