"""Synthetic waterfall source for benchmarks and demos (distribution of the reference's
``SyntheticDataGenerator._generate_single_sample``, data_generation/synthetic_generator.py:520-656)."""
from .synthetic import SyntheticWaterfalls, make_training_patches

__all__ = ["SyntheticWaterfalls", "make_training_patches"]
