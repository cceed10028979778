"""Container type of the path (reference datasets/batched_dataset.py:10-76)."""
from .batched_dataset import TorchDataset

__all__ = ["TorchDataset"]
