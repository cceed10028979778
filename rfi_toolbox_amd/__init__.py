"""rfi_toolbox_amd -- MI355X (gfx950) native implementation of the rfi_toolbox segmentation hot
path: U-Net training/inference, the Preprocessor's per-patch channel extraction and the
segmentation metrics, behind the reference's Python surface.  Compute lives in librfi_hip.so
(hand-written HIP, C ABI in include/rfi_hip.h); there is no CPU fallback."""
__version__ = "0.1.0"

# Sub-modules are imported lazily; every one of them imports ``_lib`` first, which loads
# librfi_hip.so and raises ImportError (with the build command) if it has not been built.
# ``python -m rfi_toolbox_amd.build`` must stay importable without the library.


def __getattr__(name):
    import importlib
    if name in ("models", "evaluation", "preprocessing", "datasets", "training", "distributed", "runtime",
                "data_generation"):
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
