"""Host-side mirrors of the reference's ``src/features`` hot-path modules."""
from .stft import MelSpectrogramExtractor
from .mel_sliding_window import MelAudioBuffer, MelSlidingWindowExtractor, create_mel_extractor

__all__ = ["MelSpectrogramExtractor", "MelAudioBuffer", "MelSlidingWindowExtractor", "create_mel_extractor"]
