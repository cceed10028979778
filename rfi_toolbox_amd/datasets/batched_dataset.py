"""``TorchDataset``: images (N,H,W,3) float32 + labels (N,H,W) uint8 + metadata, the container
``Preprocessor.create_dataset`` returns (reference datasets/batched_dataset.py:10-76).  Same
attributes, indexing, ``.pt`` format and dtype assertions; tensors are plain CPU tensors (the
reference additionally moves them to shared memory for DataLoader workers, which this path does
not use)."""
from __future__ import annotations

from pathlib import Path

import torch


class TorchDataset:
    def __init__(self, images, labels, metadata=None):
        assert len(images) == len(labels), "Images and labels must have same length"
        assert images.dtype == torch.float32, f"Images must be float32, got {images.dtype}"
        assert labels.dtype == torch.uint8, f"Labels must be uint8, got {labels.dtype}"
        self.images = images
        self.labels = labels
        self.metadata = metadata or {}

    def __len__(self):
        return len(self.images)

    def __getitem__(self, idx):
        return {"image": self.images[idx].contiguous(), "label": self.labels[idx].contiguous()}

    def save_to_disk(self, path):
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        torch.save({"images": self.images, "labels": self.labels, "metadata": self.metadata}, path)

    @classmethod
    def load_from_disk(cls, path):
        data = torch.load(path, weights_only=False)
        return cls(data["images"], data["labels"], data.get("metadata"))

    def __repr__(self):
        gb = (self.images.element_size() * self.images.numel()
              + self.labels.element_size() * self.labels.numel()) / 1e9
        return (f"TorchDataset(samples={len(self)}, image_shape={tuple(self.images.shape[1:])}, "
                f"size={gb:.2f}GB)")
