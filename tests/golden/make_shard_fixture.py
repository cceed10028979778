#!/usr/bin/env python3
"""Write tests/golden/ref_shards/ with the REFERENCE's own BatchWriter.

Run in the build container only (the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 CI=1 PYTHONPATH=/root/reference python tests/golden/make_shard_fixture.py

A seeded reference generator sample (64x64, 1 polarisation) goes through the reference Preprocessor
(patch_size 32 -> 16 patches in 4 views, shuffled with the global RNG) and the resulting reference
`TorchDataset` is handed to the reference `BatchWriter(samples_per_batch=5)` in two `add_batch` calls, exactly as
`SyntheticDataGenerator.generate` streams its batches to disk (synthetic_generator.py:250-261).  Output: the
shards `batch_000.pt ...`, the `metadata.json` the reference writes (with its hard-coded 1024x1024 shapes,
batched_dataset.py:168-169) and `expected.npz` = the arrays in writing order.  Files are DATA produced by
reference code; nothing of its source is stored.
"""
import os
import shutil
import sys

import numpy as np
import torch

os.environ.setdefault("CI", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    from rfi_toolbox.config.loader import DataConfig
    from rfi_toolbox.data_generation.synthetic_generator import SyntheticDataGenerator
    from rfi_toolbox.datasets.batched_dataset import BatchWriter, TorchDataset
    from rfi_toolbox.preprocessing.preprocessor import Preprocessor

    counts = {"narrowband_persistent": 2, "broadband_persistent": 1, "frequency_sweep": 1}
    cfg = DataConfig({"synthetic": {"rfi_type_counts": counts, "rfi_types": list(counts)}, "processing": {}})
    g = SyntheticDataGenerator(cfg)
    np.random.seed(4242)
    w, m, _ = g._generate_single_sample(
        num_channels=64, num_times=64, noise_level=1.0, rfi_power_min=1000.0, rfi_power_max=10000.0,
        rfi_config=g._parse_rfi_config(cfg.synthetic), enable_bandpass=True, bandpass_order=8,
        num_polarizations=1, pol_corr=0.8, synth_config=cfg.synthetic)
    np.random.seed(4243)
    ds = Preprocessor(w, flags=m).create_dataset(patch_size=32, num_workers=0)
    assert isinstance(ds, TorchDataset) and ds.images.dtype == torch.float32 and ds.labels.dtype == torch.uint8
    n = len(ds)
    out = os.path.join(HERE, "ref_shards")
    shutil.rmtree(out, ignore_errors=True)
    wr = BatchWriter(out, samples_per_batch=5)
    half = n // 2
    wr.add_batch(TorchDataset(ds.images[:half].clone(), ds.labels[:half].clone()))
    wr.add_batch(TorchDataset(ds.images[half:].clone(), ds.labels[half:].clone()))
    wr.finalize()
    np.savez_compressed(os.path.join(out, "expected.npz"), images=ds.images.numpy(), labels=ds.labels.numpy())
    for f in sorted(os.listdir(out)):
        print(f, os.path.getsize(os.path.join(out, f)))


if __name__ == "__main__":
    main()
