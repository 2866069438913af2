"""Plugin API = the drop-in boundary on the Python side (src/auralis/models/base.py:57-224, registry.py:1-3).

Same method names, argument meaning and 4-tuple return shape as the reference so that the facade (api/tts.py) and
any third-party engine written against the reference interface keep working.  The reference additionally derives
from torch.nn.Module; weights here live inside the HIP library, so `device`/`dtype` are plain properties."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Any, AsyncGenerator, Dict, List, Optional, Tuple

from .output import TTSOutput
from .requests import TTSRequest


@dataclass
class ConditioningConfig:
    speaker_embeddings: bool = False
    gpt_like_decoder_conditioning: bool = False


class BaseAsyncTTSEngine(ABC):
    @abstractmethod
    async def get_generation_context(self, request: TTSRequest) -> Tuple[List[Any], List[str], Any, Any]:
        """-> (token generators, request ids, speaker embeddings, gpt-like conditioning)"""

    @abstractmethod
    def process_tokens_to_speech(self, generator: Any, speaker_embeddings: Any, multimodal_data: Any = None,
                                 request: Optional[TTSRequest] = None) -> AsyncGenerator[TTSOutput, None]:
        """async generator of TTSOutput for one token generator"""

    @property
    def conditioning_config(self) -> ConditioningConfig:
        raise NotImplementedError

    @abstractmethod
    def get_memory_usage_curve(self):
        ...

    @classmethod
    def from_pretrained(cls, *args, **kwargs) -> "BaseAsyncTTSEngine":
        raise NotImplementedError

    async def shutdown(self) -> None:
        return None


MODEL_REGISTRY: Dict[str, type] = {}


def register_model(name: str, model_cls: type) -> None:
    MODEL_REGISTRY[name] = model_cls
