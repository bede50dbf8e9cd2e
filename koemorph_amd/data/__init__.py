from .sequential_dataset import SequentialKoeMorphDataset, detect_source_fps, load_jsonl_labels  # noqa: F401
