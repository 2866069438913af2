"""Model constants for the XTTSv2 hot path, restated as plain dataclasses.

Reference: src/auralis/models/xttsv2/config/xttsv2_gpt_config.py:133-186 (GPT dims),
src/auralis/models/xttsv2/components/tts/layers/xtts/hifigan_decoder.py:700-723 (vocoder dims),
src/auralis/models/xttsv2/XTTSv2.py:115-124 (token ids / lengths).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List


@dataclass
class GPTDims:
    hidden: int = 1024
    n_layer: int = 30
    n_head: int = 16
    head_dim: int = 64
    n_inner: int = 4096
    ln_eps: float = 1e-5
    mel_vocab: int = 1026           # num_audio_tokens
    start_token: int = 1024         # start_audio_token
    stop_token: int = 1025          # stop_audio_token
    mel_positions: int = 608        # max_audio_tokens + 3 (vllm_mm_gpt.py:753)
    max_audio_tokens: int = 605
    text_vocab: int = 6681
    text_positions: int = 404       # max_text_tokens + 2
    perceiver_latents: int = 32
    max_model_len: int = 1047       # XTTSv2.py:217-219
    activation: str = "gelu_new"    # checkpoint_converter.py:197


@dataclass
class VocoderDims:
    in_dim: int = 1024
    initial_channel: int = 512
    upsample_rates: List[int] = field(default_factory=lambda: [8, 8, 2, 2])
    upsample_kernels: List[int] = field(default_factory=lambda: [16, 16, 4, 4])
    resblock_kernels: List[int] = field(default_factory=lambda: [3, 7, 11])
    resblock_dilations: List[int] = field(default_factory=lambda: [1, 3, 5])
    d_vector: int = 512
    input_sample_rate: int = 22050
    output_sample_rate: int = 24000
    output_hop_length: int = 256
    ar_mel_length_compression: int = 1024
    lrelu_slope: float = 0.1
    post_lrelu_slope: float = 0.01  # F.leaky_relu default, hifigan_decoder.py:257

    def stage_channels(self) -> List[int]:
        return [self.initial_channel // (2 ** (i + 1)) for i in range(len(self.upsample_rates))]

    def frames_for_latents(self, n_latent: int) -> int:
        """T' = floor(floor(4 T) * 24000/22050) (hifigan_decoder.py:787-800)."""
        import math
        s1 = self.ar_mel_length_compression / self.output_hop_length
        s2 = self.output_sample_rate / self.input_sample_rate
        return int(math.floor(int(math.floor(n_latent * s1)) * s2))

    def samples_for_latents(self, n_latent: int) -> int:
        n = self.frames_for_latents(n_latent)
        for r in self.upsample_rates:
            n *= r
        return n


@dataclass
class XTTSDims:
    gpt: GPTDims = field(default_factory=GPTDims)
    voc: VocoderDims = field(default_factory=VocoderDims)
    sample_rate: int = 24000
