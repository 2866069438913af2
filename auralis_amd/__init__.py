"""auralis_amd — MI355X-native XTTSv2 generate_speech() path behind the Auralis API.

    from auralis_amd import TTS, TTSRequest, TTSOutput
"""
from .api.engine_base import MODEL_REGISTRY, BaseAsyncTTSEngine, ConditioningConfig, register_model  # noqa: F401
from .api.output import TTSOutput  # noqa: F401
from .api.requests import AudioPreprocessingConfig, TTSRequest  # noqa: F401
from .api.tts import TTS  # noqa: F401

__version__ = "0.1.0"
