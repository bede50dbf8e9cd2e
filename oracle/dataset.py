"""CPU restatement of the reference's sequential window producer (test infrastructure, like the rest of oracle/).

Follows src/data/sequential_dataset.py: label resampling :136-154 (np.linspace + np.interp -- numpy IS the library
the reference calls, so this oracle is pinned by construction), alignment :166-178, window slicing :180-206."""
import numpy as np


def resample_blendshapes(blendshapes: np.ndarray, source_fps: float, target_fps: float) -> np.ndarray:
    if abs(source_fps - target_fps) < 0.1:
        return blendshapes
    ratio = target_fps / source_fps
    source_len = len(blendshapes)
    target_len = int(source_len * ratio)
    source_indices = np.linspace(0, source_len - 1, target_len)
    resampled = np.zeros((target_len, blendshapes.shape[1]), dtype=np.float32)
    for i in range(blendshapes.shape[1]):
        resampled[:, i] = np.interp(source_indices, np.arange(source_len), blendshapes[:, i])
    return resampled


def windows(audio: np.ndarray, blendshapes: np.ndarray, window_frames=256, stride_frames=1, hop_length=533):
    """Yields (window index, start frame, audio window, label window) exactly as _process_file_pair does."""
    expected_frames = len(audio) // hop_length
    if abs(len(blendshapes) - expected_frames) > 1:
        num_frames = min(len(blendshapes), expected_frames)
        audio = audio[:num_frames * hop_length]
        blendshapes = blendshapes[:num_frames]
    num_windows = (len(blendshapes) - window_frames) // stride_frames + 1
    window_samples = window_frames * hop_length
    for i in range(num_windows):
        start_frame = i * stride_frames
        a = audio[start_frame * hop_length:(start_frame + window_frames) * hop_length]
        b = blendshapes[start_frame:start_frame + window_frames]
        if len(a) == window_samples and len(b) == window_frames:
            yield i, start_frame, a, b
