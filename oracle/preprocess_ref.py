"""NumPy oracle for ``Preprocessor.create_dataset``.  TEST INFRASTRUCTURE ONLY.

Restates rfi_toolbox/preprocessing/preprocessor.py (reference v0.2.0):

* tiling            ``patchify``/``_create_patches``            :22-42, :478-560
* views             ``_apply_rotations``                        :413-446
* complex channels  ``_extract_channels_from_complex``          :562-606
* real channels     ``_extract_channels_from_real``             :608-644
* median normalise  ``_normalize``                              :646-670
* stretch           ``_apply_stretch``                          :672-706
* MAD flags         ``_generate_mad_flags``                     :708-745
* blank removal     ``_remove_blank_patches``                   :746-756
* shuffle           ``_shuffle`` (global ``np.random``)         :758-763
* ImageNet norm     ``_apply_sam2_normalization``               :765-783
* pipeline order    ``create_dataset``                          :198-411

Written as array-at-once NumPy (reshape/transpose tiling, vectorised channel
math over the whole patch stack) rather than the reference's per-patch Python
loops; the results are pinned to the reference by tests/golden.
"""
from __future__ import annotations

import numpy as np

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)
LOG_LO, LOG_HI = -3.0, 4.0


def tile2d(a, ps):
    """(H,W) -> (H//ps * W//ps, ps, ps) row-major tiles; zero-pads up to a multiple of ps
    (also when a side is shorter than ps).  == patchify(step=ps) after the reference's padding."""
    h, w = a.shape
    ph = (-h) % ps if h >= ps else ps - h
    pw = (-w) % ps if w >= ps else ps - w
    if ph or pw:
        a = np.pad(a, ((0, ph), (0, pw)), mode="constant", constant_values=0)
    H, W = a.shape
    return a.reshape(H // ps, ps, W // ps, ps).transpose(0, 2, 1, 3).reshape(-1, ps, ps)


def views(pol, rotations):
    """[orig, flipud, T, flipud(T)] truncated to ``rotations`` in {1,2,4}."""
    out = [pol]
    if rotations >= 2:
        out.append(pol[::-1, :])
    if rotations >= 4:
        out.append(pol.T)
        out.append(pol.T[::-1, :])
    return out


def expand(data4, rotations, augment=True):
    """(B,P,C,T) -> list of 2-D waterfalls in the reference's order."""
    r = rotations if (augment and rotations > 1) else 1
    out = []
    for bl in data4:
        for pol in bl:
            out.extend(views(pol, r))
    return out


def to_patches(waterfalls, ps):
    """List of 2-D arrays -> stacked patches; whole waterfalls when they already fit."""
    h0, w0 = waterfalls[0].shape
    if h0 <= ps and w0 <= ps:
        return np.array(waterfalls)
    return np.concatenate([tile2d(np.asarray(w), ps) for w in waterfalls], axis=0)


def _minmax(stack):
    lo = np.nanmin(stack, axis=(1, 2), keepdims=True)
    hi = np.nanmax(stack, axis=(1, 2), keepdims=True)
    span = hi - lo
    ok = span > 0
    return np.where(ok, (stack - lo) / np.where(ok, span, 1.0), 0.0)


def _grad_mag(log_amp):
    d0 = np.zeros_like(log_amp)
    d1 = np.zeros_like(log_amp)
    d0[:, 1:, :] = log_amp[:, 1:, :] - log_amp[:, :-1, :]
    d1[:, :, 1:] = log_amp[:, :, 1:] - log_amp[:, :, :-1]
    return np.sqrt(d0 * d0 + d1 * d1)


def channels_complex(z):
    """(N,H,W) complex -> (N,H,W,3) float64 [gradient, log-amp, phase] each in [0,1]."""
    log_amp = np.log10(np.abs(z) + 1e-10)
    ch0 = _minmax(_grad_mag(log_amp))
    ch1 = np.clip((log_amp - LOG_LO) / (LOG_HI - LOG_LO), 0, 1)
    ch2 = (np.angle(z) + np.pi) / (2 * np.pi)
    return np.stack([ch0, ch1, ch2], axis=-1)


def channels_real(x):
    log_amp = np.log10(np.abs(x) + 1e-10)
    ch0 = _minmax(_grad_mag(log_amp))
    ch1 = _minmax(log_amp)
    return np.stack([ch0, ch1, np.zeros_like(log_amp)], axis=-1)


def median_normalise(stack):
    med = np.nanmedian(stack.reshape(len(stack), -1), axis=1)
    scale = np.where(med > 0, med, 1.0)
    return stack / scale[:, None, None]


def _mad(v):
    v = v[~np.isnan(v)]
    return np.median(np.abs(v - np.median(v)))


def stretch(stack, kind):
    if kind == "SQRT":
        fn = np.sqrt
    elif kind == "LOG10":
        fn = np.log10
    else:
        raise ValueError(f"Invalid stretch '{kind}'. Use 'SQRT' or 'LOG10'")
    out = []
    with np.errstate(divide="ignore", invalid="ignore"):
        for p in stack:
            s = fn(np.abs(p))
            finite = s[np.isfinite(s)]
            s[np.isinf(s)] = _mad(finite) if finite.size else 0
            out.append(s)
    return np.array(out)


def mad_flags(stack, sigma):
    out = np.zeros(stack.shape, dtype=bool)
    for i, p in enumerate(stack):
        med = np.nanmedian(p)
        mad = _mad(p.ravel())
        out[i] = (p > med + mad * sigma) | (p < med - mad * sigma)
    return out


def create_dataset(data, flags=None, patch_size=128, stretch_kind=None, flag_sigma=5,
                   use_custom_flags=True, num_patches=None, normalize_before_stretch=True,
                   normalize_after_stretch=False, enable_augmentation=True,
                   augmentation_rotations=4, inference_mode=False, rng_permutation=None):
    """Returns (images f32 (N,ps,ps,3) ImageNet-normalised, labels u8 (N,ps,ps)).

    ``rng_permutation``: callable n -> permutation (defaults to the global
    ``np.random.permutation`` exactly like the reference's ``_shuffle``)."""
    data = np.asarray(data)
    if data.ndim == 3:
        data = data[None]
    elif data.ndim != 4:
        raise ValueError(f"Data must be 3D or 4D, got shape {data.shape}")
    have_flags = use_custom_flags and flags is not None
    wf = expand(data, augmentation_rotations, enable_augmentation)
    patches = to_patches(wf, patch_size)
    pflags = None
    if have_flags:
        fl = np.asarray(flags)
        if fl.ndim == 3:
            fl = fl[None]
        pflags = to_patches(expand(fl, augmentation_rotations, enable_augmentation), patch_size)

    if not np.iscomplexobj(patches):
        if normalize_before_stretch:
            patches = median_normalise(patches)
        if stretch_kind:
            patches = stretch(patches, stretch_kind)
        if normalize_after_stretch:
            patches = median_normalise(patches)

    if inference_mode:
        pflags = np.zeros(patches.shape, dtype=np.uint8)
    elif pflags is None:
        pflags = mad_flags(patches, flag_sigma)

    if not inference_mode:
        keep = pflags.reshape(len(pflags), -1).any(axis=1)
        if keep.any():
            patches, pflags = patches[keep], pflags[keep]
        perm = (rng_permutation or np.random.permutation)(len(patches))
        patches, pflags = patches[perm], pflags[perm]
    if num_patches and num_patches < len(patches):
        patches, pflags = patches[:num_patches], pflags[:num_patches]

    ch = channels_complex(patches) if np.iscomplexobj(patches) else channels_real(patches)
    images = (ch.astype(np.float32) - IMAGENET_MEAN) / IMAGENET_STD
    return images.astype(np.float32), np.asarray(pflags).astype(np.uint8)
