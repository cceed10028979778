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


def _shard_files(directory):
    import json
    d = Path(directory)
    # by shard NUMBER: the writer names them batch_{idx:03d}.pt (reference datasets/batched_dataset.py:147), so from the
    # 1000th shard on the lexicographic order (batch_1000 < batch_101) is not the written order
    def shard_no(p):
        digits = "".join(ch for ch in p.stem[len("batch_"):] if ch.isdigit())
        return (int(digits) if digits else -1, p.name)
    files = sorted(d.glob("batch_*.pt"), key=shard_no)
    if not files:
        raise FileNotFoundError(f"no batch_*.pt shards under {d}")
    meta = json.load(open(d / "metadata.json")) if (d / "metadata.json").exists() else {}
    return files, meta


def _check_shard(p, f):
    img, lab = p["images"], p["labels"]
    if img.ndim != 4 or img.shape[-1] != 3 or lab.ndim != 3 or tuple(lab.shape) != tuple(img.shape[:3]):
        raise ValueError(f"{f}: expected images (n,H,W,3) and labels (n,H,W), got {tuple(img.shape)} / {tuple(lab.shape)}")
    return img, lab


def load_batches(directory):
    """Read every ``batch_*.pt`` shard a (reference or local) BatchWriter wrote into one TorchDataset.

    Shapes come from the tensors: the reference's ``metadata.json`` hard-codes ``image_shape [1024,1024,3]``
    whatever was written (datasets/batched_dataset.py:168-169), so it is kept as metadata only.  A reference shard
    is a VIEW into the whole flushed block (``torch.save`` of a slice stores the full storage); loading gives the
    slice back."""
    files, meta = _shard_files(directory)
    parts = [_check_shard(torch.load(f, weights_only=True), f) for f in files]      # tensors only: no pickle code from the data directory
    return TorchDataset(torch.cat([i for i, _ in parts]).float(),
                        torch.cat([lb for _, lb in parts]).to(torch.uint8), meta)


def load_batches_device(directory, device=None):
    """The same shards straight into HBM: each ``batch_*.pt`` is memory-mapped and its patches are copied from the
    mapping into their slot of ONE (N,H,W,3) float32 / (N,H,W) uint8 device buffer pair -- the NHWC layout the
    kernels consume -- without a concatenated host copy.  -> (images DeviceArray, labels DeviceArray, metadata)."""
    import ctypes as C

    import numpy as np

    from .._lib import DEVICE, HOST, check, lib
    from ..runtime import Context
    files, meta = _shard_files(directory)
    parts = [_check_shard(torch.load(f, mmap=True, weights_only=True), f) for f in files]
    shape = tuple(parts[0][0].shape[1:])
    for (img, _), f in zip(parts, files):
        if tuple(img.shape[1:]) != shape:
            raise ValueError(f"{f}: patch shape {tuple(img.shape[1:])} differs from {shape}")
    n = sum(len(i) for i, _ in parts)
    ctx = Context.get(device)
    images, labels = ctx.empty((n, *shape), np.float32), ctx.empty((n, *shape[:2]), np.uint8)
    off = 0
    for img, lab in parts:
        img, lab = img.to(torch.float32).contiguous(), lab.to(torch.uint8).contiguous()     # no-ops for reference shards
        k = len(img)
        per = shape[0] * shape[1]
        check(lib.rfi_memcpy(ctx.handle, C.c_void_p(images.ptr + off * per * 3 * 4), DEVICE,
                             C.c_void_p(img.data_ptr()), HOST, k * per * 3 * 4))
        check(lib.rfi_memcpy(ctx.handle, C.c_void_p(labels.ptr + off * per), DEVICE,
                             C.c_void_p(lab.data_ptr()), HOST, k * per))
        off += k
    return images, labels, meta
