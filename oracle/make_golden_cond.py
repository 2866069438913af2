"""Generate tests/golden/cond_female_6s.npz: speaker conditioning of real speech through the REFERENCE's own module classes.

TEST INFRASTRUCTURE ONLY.  Usage (build container): python -m oracle.make_golden_cond
Input: the first 6 s of the reference's test resource tests/resources/audio_samples/female.wav, loaded and resampled to
22 050 Hz by auralis_amd/conditioning.py (stored in the fixture as int16 so that the GPU box, which has no /root/reference,
can run the same clip).  What runs from /root/reference, unmodified (oracle/ref_import.py): ConditioningEncoder
(latent_encoder.py:134-253), PerceiverResampler (perceiver_encoder.py:363-442) and ResNetSpeakerEncoder
(hifigan_decoder.py:602-646, with its torchaudio spectrogram switched off: torchaudio is not installed, so the mel
front-end is OURS in both legs and stays "parity unpinned"; it is cross-checked independently in tests/test_conditioning.py).
Weights: the seeded synthetic conditioning weights (seed 99).  Stored: the clip, the two mels the front-end produced, and
the reference classes' gpt_cond_latent [1,32,1024] / speaker_embedding [1,512,1] on those mels.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auralis_amd import conditioning as Cn  # noqa: E402
from auralis_amd.checkpoint import make_synthetic_conditioning_weights  # noqa: E402
from auralis_amd.config import XTTSDims  # noqa: E402
from oracle.ref_import import load_reference_hifigan, load_reference_xtts_layer  # noqa: E402

WAV = "/root/reference/tests/resources/audio_samples/female.wav"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cond_female_6s.npz")


@torch.no_grad()
def main():
    dims = XTTSDims()
    sd = make_synthetic_conditioning_weights(dims, seed=99)
    sd["mel_stats"] = torch.ones(80)
    audio = Cn.load_audio(WAV, 22050)[:, : 22050 * 6]
    pcm16 = torch.round(audio.clamp(-1, 1) * 32767.0).to(torch.int16)
    audio = pcm16.float() / 32767.0                       # the fixture's clip is what both legs see
    # --- front-end (ours) ---
    mel_gpt = Cn.mel_spectrogram(audio, 22050, 2048, 1024, 256, 80, 0.0, 8000.0, "hann", slaney_norm=True)
    mel_gpt = torch.log(torch.clamp(mel_gpt, min=1e-5)) / sd["mel_stats"][None, :, None]
    a16 = Cn.resample(audio, 22050, 16000)
    x = torch.nn.functional.pad(a16.unsqueeze(1), (1, 0), "reflect")
    x = torch.nn.functional.conv1d(x, torch.tensor([-0.97, 1.0]).view(1, 1, -1)).squeeze(1)
    mel_spk = Cn.mel_spectrogram(x, 16000, 512, 400, 160, 64, window="hamming", slaney_norm=False)
    # --- network bodies: the reference's classes ---
    le, pe, hg = load_reference_xtts_layer("latent_encoder"), load_reference_xtts_layer("perceiver_encoder"), load_reference_hifigan()
    enc = le.ConditioningEncoder(80, 1024, num_attn_heads=16).eval()
    enc.load_state_dict({k[len("conditioning_encoder."):]: v for k, v in sd.items() if k.startswith("conditioning_encoder.")})
    per = pe.PerceiverResampler(dim=1024, depth=2, dim_context=1024, num_latents=32, dim_head=64, heads=8, ff_mult=4).eval()
    per.load_state_dict({k[len("conditioning_perceiver."):]: v for k, v in sd.items() if k.startswith("conditioning_perceiver.")})
    se = hg.HifiDecoder().speaker_encoder.eval()
    se.load_state_dict({k[len("hifigan_decoder.speaker_encoder."):]: v for k, v in sd.items()
                        if k.startswith("hifigan_decoder.speaker_encoder.")})
    se.use_torch_spec = False
    latent = per(enc(mel_gpt).permute(0, 2, 1))           # one 6-s chunk -> the mean over chunks is the chunk itself
    emb = se(mel_spk.clone(), l2_norm=True).unsqueeze(-1)
    ours_l, ours_e = Cn.get_conditioning_latents(sd, [_wav_bytes(pcm16)], max_ref_length=30, gpt_cond_len=6, gpt_cond_chunk_len=6)
    print("latent", tuple(latent.shape), "max |ours - reference classes|", float((ours_l - latent).abs().max()),
          "| embedding", float((ours_e - emb).abs().max()))
    np.savez_compressed(OUT, pcm16=pcm16.numpy()[0], mel_gpt=mel_gpt.numpy().astype(np.float16), mel_spk_log=torch.log(mel_spk + 1e-6).numpy().astype(np.float16),
                        gpt_cond_latent=latent.numpy().astype(np.float32), speaker_embedding=emb.numpy().astype(np.float32))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


def _wav_bytes(pcm16: torch.Tensor) -> bytes:
    import io
    import wave
    b = io.BytesIO()
    with wave.open(b, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(22050)
        w.writeframes(pcm16.numpy()[0].tobytes())
    return b.getvalue()


if __name__ == "__main__":
    main()
