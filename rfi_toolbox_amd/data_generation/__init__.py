"""Synthetic waterfall source (distribution of the reference's
``SyntheticDataGenerator._generate_single_sample``, data_generation/synthetic_generator.py:520-815):
host NumPy back end and the on-device generator of librfi_hip.so."""
from .synthetic import SyntheticWaterfalls, make_training_patches, make_training_patches_device

__all__ = ["SyntheticWaterfalls", "make_training_patches", "make_training_patches_device"]
