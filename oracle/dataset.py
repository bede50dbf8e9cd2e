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


# ---- AdaptiveSequentialDataset (src/data/adaptive_sequential_dataset.py) ------------------------------------------------
def adaptive_stride(stride_mode, initial_stride, final_stride, epoch, max_epochs):
    """_calculate_stride :111-126."""
    if stride_mode == "dense":
        return 1
    elif stride_mode == "sparse":
        return initial_stride
    elif stride_mode == "progressive":
        progress = min(1.0, epoch / max(1, max_epochs - 1))
        stride = int(initial_stride - progress * (initial_stride - final_stride))
        return max(final_stride, stride)
    elif stride_mode == "mixed":
        return initial_stride
    raise ValueError(f"Unknown stride mode: {stride_mode}")


def adaptive_windows(audio, blendshapes, stride_mode, current_stride, initial_stride, dense_sampling_ratio=0.1,
                     window_frames=256, hop_length=533):
    """Yields (window index, start frame, is_dense, audio window, label window) in the order of _process_file_pair
    :217-283 (alignment :224-229; dense :156-180; sparse :182-209; mixed :240-278 with numpy's global generator)."""
    expected_frames = len(audio) // hop_length
    if abs(len(blendshapes) - expected_frames) > 1:
        num_frames = min(len(blendshapes), expected_frames)
        audio = audio[:num_frames * hop_length]
        blendshapes = blendshapes[:num_frames]
    window_samples = window_frames * hop_length

    def cut(start_frame):
        a = audio[start_frame * hop_length:(start_frame + window_frames) * hop_length]
        b = blendshapes[start_frame:start_frame + window_frames]
        return a, b

    def sparse(stride):
        num_windows = (len(blendshapes) - window_frames) // stride + 1
        for i in range(num_windows):
            start_frame = i * stride
            if start_frame + window_frames > len(blendshapes):
                break
            a, b = cut(start_frame)
            if len(a) == window_samples and len(b) == window_frames:
                yield i, start_frame, False, a, b

    if stride_mode == "dense":
        for i in range(len(blendshapes) - window_frames + 1):
            a, b = cut(i)
            if len(a) == window_samples and len(b) == window_frames:
                yield i, i, True, a, b
    elif stride_mode in ("sparse", "progressive"):
        yield from sparse(current_stride)
    elif stride_mode == "mixed":
        dense_samples = int((len(blendshapes) - window_frames) * dense_sampling_ratio)
        try:
            dense_indices = np.random.choice(len(blendshapes) - window_frames, size=dense_samples, replace=False)
        except Exception:       # a clip shorter than the window: the reference logs the error and yields nothing (:281-283)
            return
        for idx in sorted(dense_indices):
            a, b = cut(idx)
            if len(a) == window_samples:
                yield int(idx), int(idx), True, a, b
        yield from sparse(initial_stride)
