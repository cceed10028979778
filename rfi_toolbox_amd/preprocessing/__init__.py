"""Drop-in for ``rfi_toolbox.preprocessing`` (reference preprocessing/preprocessor.py)."""
from .preprocessor import Preprocessor, patchify

__all__ = ["Preprocessor", "patchify"]
