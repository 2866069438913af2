"""TTSRequest — same fields, defaults and meaning as the reference dataclass
(src/auralis/common/definitions/requests.py:134-277).  Fields the reference carries but never consumes on the
synthesis path (do_sample, length_penalty, sound_norm_refs, load_sample_rate) are kept for
drop-in compatibility.  `seed` is new surface (per-request noise stream; the reference has none)."""
from __future__ import annotations

import copy as _copy
import uuid
from dataclasses import dataclass, field
from typing import AsyncGenerator, Callable, List, Optional, Union

from .lang import get_language, validate_language


@dataclass
class AudioPreprocessingConfig:
    """Options of the optional reference-audio enhancer (requests.py:171, enhancer.py:34-153): carried, unused here."""
    sample_rate: int = 22050
    normalize: bool = True
    trim_silence: bool = True
    remove_noise: bool = True
    enhance_speech: bool = True


@dataclass
class TTSRequest:
    text: Union[AsyncGenerator[str, None], str, List[str]]
    speaker_files: Union[str, List[str], bytes, List[bytes]]
    context_partial_function: Optional[Callable] = None
    start_time: Optional[float] = None
    enhance_speech: bool = False
    audio_config: AudioPreprocessingConfig = field(default_factory=AudioPreprocessingConfig)
    language: str = "auto"
    request_id: str = field(default_factory=lambda: uuid.uuid4().hex)
    load_sample_rate: int = 22050
    sound_norm_refs: bool = False
    # voice conditioning
    max_ref_length: int = 60
    gpt_cond_len: int = 30
    gpt_cond_chunk_len: int = 4
    # generation
    stream: bool = False
    temperature: float = 0.75
    top_p: float = 0.85
    top_k: int = 50
    repetition_penalty: float = 5.0
    length_penalty: float = 1.0
    do_sample: bool = True
    seed: Optional[int] = None
    # new surface, like `seed`: > 0 marks the request latency-critical -- its FIRST chunk is what a listener waits for -- and is
    # handed to the engine as aur_seq_desc.priority (the reference has no such knob: tests/integration/stream_ttfb.py only measures)
    priority: int = 0

    def __post_init__(self):
        if self.language == "auto" and isinstance(self.text, str) and len(self.text) > 0:
            self.language = get_language(self.text)
        validate_language(self.language)

    def infer_language(self) -> None:
        """Detect the language of `text` when it is "auto" (the reference's requests.py method of the same name)."""
        if self.language == "auto" and isinstance(self.text, str) and len(self.text) > 0:
            self.language = get_language(self.text)

    def copy(self) -> "TTSRequest":
        return _copy.copy(self)
