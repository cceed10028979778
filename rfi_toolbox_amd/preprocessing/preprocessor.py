"""``Preprocessor`` with the reference's constructor and ``create_dataset`` signature
(rfi_toolbox/preprocessing/preprocessor.py:175-211).

Split of work:
* host (NumPy views, no per-pixel arithmetic): the 4-way views (:413-446), zero-padded tiling
  (:478-560, ``patchify`` :22-42), blank-patch removal (:746-756), the global-RNG shuffle
  (:758-763) and ``num_patches`` truncation -- index bookkeeping over whole patches;
  for REAL input also the median-normalise / stretch / MAD-flag steps (:646-745), which need
  order statistics and are outside this round's device scope.
* MI355X (librfi_hip.so ``rfi_preprocess_patches``): the hot loop (:366-384) -- per patch
  log-amplitude, phase, forward-difference gradient magnitude with per-patch min-max,
  fixed-range log-amp scaling, float32 cast and ImageNet normalisation -> NHWC float32.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .._lib import C128, C64, DEVICE, F32, F64, HOST, check, lib
from ..datasets.batched_dataset import TorchDataset
from ..runtime import Context


def patchify(array, patch_shape, step):
    """(H,W) -> (n_h, n_w, ph, pw) windows, ``step`` apart (reference :22-42; no padding here)."""
    a = np.asarray(array)
    ph, pw = patch_shape
    nh = (a.shape[0] - ph) // step + 1
    nw = (a.shape[1] - pw) // step + 1
    s0, s1 = a.strides
    v = np.lib.stride_tricks.as_strided(a, shape=(nh, nw, ph, pw), strides=(s0 * step, s1 * step, s0, s1),
                                        writeable=False)
    return np.ascontiguousarray(v)


def _tile(a, ps):
    h, w = a.shape
    ph = (ps - h) if h < ps else (-h) % ps
    pw = (ps - w) if w < ps else (-w) % ps
    if ph or pw:
        a = np.pad(a, ((0, ph), (0, pw)), mode="constant", constant_values=0)
    return patchify(a, (ps, ps), ps).reshape(-1, ps, ps)


def _views(data4, rotations):
    out = []
    for bl in data4:
        for pol in bl:
            out.append(pol)
            if rotations >= 2:
                out.append(pol[::-1, :])
            if rotations >= 4:
                out.append(pol.T)
                out.append(pol.T[::-1, :])
    return out


def _to_patches(wfs, ps):
    h0, w0 = wfs[0].shape
    if h0 <= ps and w0 <= ps:
        return np.array(wfs), None
    return np.concatenate([_tile(np.asarray(w), ps) for w in wfs], axis=0), [w.shape for w in wfs]


def _mad(v):
    v = v[~np.isnan(v)]
    return np.median(np.abs(v - np.median(v)))


class Preprocessor:
    def __init__(self, data, flags=None, device=None):
        data = np.asarray(data)
        if data.ndim == 4:
            self.data = data
        elif data.ndim == 3:
            self.data = data[np.newaxis, ...]
        else:
            raise ValueError(f"Data must be 3D or 4D, got shape {data.shape}")
        self.flags = flags
        self.patches = None
        self.patch_flags = None
        self.dataset = None
        self._device = device

    # ---- device hot loop
    def _channels_on_device(self, patches):
        n, ph, pw = patches.shape
        code = {np.dtype(np.complex128): C128, np.dtype(np.complex64): C64, np.dtype(np.float64): F64,
                np.dtype(np.float32): F32}.get(patches.dtype)
        if code is None:
            patches = patches.astype(np.complex128 if np.iscomplexobj(patches) else np.float64)
            code = C128 if np.iscomplexobj(patches) else F64
        patches = np.ascontiguousarray(patches)
        out = np.empty((n, ph, pw, 3), dtype=np.float32)
        if n:
            ctx = Context.get(self._device)
            check(lib.rfi_preprocess_patches(ctx.handle, patches.ctypes.data_as(C.c_void_p), HOST, code, n, ph,
                                             pw, out.ctypes.data_as(C.c_void_p), HOST))
        return out

    def create_dataset(self, patch_size=128, stretch=None, flag_sigma=5, use_custom_flags=True,
                       num_patches=None, normalize_before_stretch=True, normalize_after_stretch=False,
                       num_workers=4, enable_augmentation=True, augmentation_rotations=4,
                       inference_mode=False):
        del num_workers                       # the hot loop runs on the GPU, no worker pool
        rot = augmentation_rotations if (enable_augmentation and augmentation_rotations > 1) else 1
        have_flags = use_custom_flags and self.flags is not None
        patches, shapes = _to_patches(_views(self.data, rot), patch_size)
        if shapes is not None:
            self.original_shapes = shapes
        pflags = None
        if have_flags:
            fl = np.asarray(self.flags)
            fl = fl[np.newaxis, ...] if fl.ndim == 3 else fl
            pflags, _ = _to_patches(_views(fl, rot), patch_size)
        if stretch not in (None, "", "SQRT", "LOG10") and not np.iscomplexobj(patches):
            raise ValueError(f"Invalid stretch '{stretch}'. Use 'SQRT' or 'LOG10'")
        if not np.iscomplexobj(patches):
            patches = self._real_pipeline(patches, stretch, normalize_before_stretch, normalize_after_stretch)
        if inference_mode:
            pflags = np.zeros(patches.shape, dtype=np.uint8)
        elif pflags is None:
            pflags = self._mad_flags(patches, flag_sigma)
        if not inference_mode:
            keep = pflags.reshape(len(pflags), -1).any(axis=1)
            if keep.any():
                patches, pflags = patches[keep], pflags[keep]
            perm = np.random.permutation(len(patches))         # global RNG, as the reference (:760)
            patches, pflags = patches[perm], pflags[perm]
        if num_patches and num_patches < len(patches):
            patches, pflags = patches[:num_patches], pflags[:num_patches]
        self.patches, self.patch_flags = patches, pflags
        images = self._channels_on_device(patches)
        labels = np.ascontiguousarray(pflags).astype(np.uint8)
        metadata = {"patch_size": patch_size, "stretch": stretch, "flag_sigma": flag_sigma,
                    "normalize_before_stretch": normalize_before_stretch,
                    "normalize_after_stretch": normalize_after_stretch,
                    "augmentation_rotations": augmentation_rotations,
                    "original_shapes": getattr(self, "original_shapes", None)}
        self.dataset = TorchDataset(torch.from_numpy(images), torch.from_numpy(labels), metadata)
        return self.dataset

    # ---- real-valued branch (host, order statistics)
    @staticmethod
    def _median_normalise(stack):
        med = np.nanmedian(stack.reshape(len(stack), -1), axis=1)
        return stack / np.where(med > 0, med, 1.0)[:, None, None]

    def _real_pipeline(self, patches, stretch, before, after):
        if before:
            patches = self._median_normalise(patches)
        if stretch:
            fn = np.sqrt if stretch == "SQRT" else np.log10
            out = []
            with np.errstate(divide="ignore", invalid="ignore"):
                for p in patches:
                    s = fn(np.abs(p))
                    finite = s[np.isfinite(s)]
                    s[np.isinf(s)] = _mad(finite) if finite.size else 0
                    out.append(s)
            patches = np.array(out)
        if after:
            patches = self._median_normalise(patches)
        return patches

    @staticmethod
    def _mad_flags(patches, sigma):
        out = np.zeros(patches.shape, dtype=bool)
        for i, p in enumerate(patches):
            p = np.abs(p) if np.iscomplexobj(p) else p
            med, mad = np.nanmedian(p), _mad(p.ravel())
            out[i] = (p > med + mad * sigma) | (p < med - mad * sigma)
        return out
