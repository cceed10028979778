"""Container type of the path (reference datasets/batched_dataset.py:10-76)."""
from .batched_dataset import BatchWriter, TorchDataset, load_batches, load_batches_device

__all__ = ["TorchDataset", "BatchWriter", "load_batches", "load_batches_device"]
