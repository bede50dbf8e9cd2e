"""Host-side mirrors of the reference's ``src/features`` hot-path modules."""
from .stft import MelSpectrogramExtractor
from .mel_sliding_window import MelAudioBuffer, MelSlidingWindowExtractor, create_mel_extractor

__all__ = ["MelSpectrogramExtractor", "MelAudioBuffer", "MelSlidingWindowExtractor", "create_mel_extractor"]
from .opensmile_extractor import AudioBuffer, OpenSMILEeGeMAPSExtractor, create_opensmile_extractor  # noqa: E402

__all__ += ["AudioBuffer", "OpenSMILEeGeMAPSExtractor", "create_opensmile_extractor"]
