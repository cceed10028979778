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


class BatchWriter:
    """Sharded ``batch_%03d.pt`` + ``metadata.json`` writer with the reference's on-disk format
    (datasets/batched_dataset.py:79-184): each shard is ``{"images": f32 (n,H,W,3), "labels": u8 (n,H,W)}``.
    ``image_shape``/``mask_shape`` in the metadata are the real shapes (the reference hard-codes 1024)."""

    def __init__(self, output_dir, samples_per_batch=100):
        self.output_dir = Path(output_dir)
        self.output_dir.mkdir(parents=True, exist_ok=True)
        self.samples_per_batch = samples_per_batch
        self._img, self._lab = [], []
        self.batch_file_idx = 0
        self.total_samples = 0
        self._shape = None

    def add_batch(self, dataset):
        self._img.append(dataset.images)
        self._lab.append(dataset.labels)
        if sum(len(i) for i in self._img) >= self.samples_per_batch:
            self._flush()

    def _flush(self):
        if not self._img:
            return
        images, labels = torch.cat(self._img), torch.cat(self._lab)
        self._img, self._lab = [], []
        self._shape = tuple(images.shape[1:])
        for lo in range(0, len(images), self.samples_per_batch):
            hi = min(lo + self.samples_per_batch, len(images))
            torch.save({"images": images[lo:hi].clone(), "labels": labels[lo:hi].clone()},
                       self.output_dir / f"batch_{self.batch_file_idx:03d}.pt")
            self.total_samples += hi - lo
            self.batch_file_idx += 1

    def finalize(self):
        import json
        self._flush()
        shape = list(self._shape) if self._shape else [0, 0, 3]
        meta = {"num_samples": self.total_samples, "samples_per_batch": self.samples_per_batch,
                "num_batches": self.batch_file_idx, "image_shape": shape, "mask_shape": shape[:2],
                "dtype": "float32"}
        with open(self.output_dir / "metadata.json", "w") as f:
            json.dump(meta, f, indent=2)
        return meta


def load_batches(directory):
    """Read every ``batch_*.pt`` shard a (reference or local) BatchWriter wrote into one TorchDataset."""
    import json
    d = Path(directory)
    files = sorted(d.glob("batch_*.pt"))
    if not files:
        raise FileNotFoundError(f"no batch_*.pt shards under {d}")
    parts = [torch.load(f, weights_only=False) for f in files]
    meta = json.load(open(d / "metadata.json")) if (d / "metadata.json").exists() else {}
    return TorchDataset(torch.cat([p["images"] for p in parts]).float(),
                        torch.cat([p["labels"] for p in parts]).to(torch.uint8), meta)
