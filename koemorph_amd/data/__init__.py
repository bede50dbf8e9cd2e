from .sequential_dataset import SequentialKoeMorphDataset, detect_source_fps, load_jsonl_labels  # noqa: F401
from .adaptive_sequential_dataset import (AdaptiveSequentialDataset, calculate_stride, create_adaptive_dataloader,  # noqa: F401
                                          window_plan)
