"""``Preprocessor`` with the reference's constructor and ``create_dataset`` signature
(rfi_toolbox/preprocessing/preprocessor.py:175-211).

Split of work for the training path (complex visibilities + caller-supplied flags):
* MI355X, from the waterfall itself (librfi_hip.so ``rfi_patch_any_flag`` +
  ``rfi_preprocess_gather``): the 4-way views (:413-446), the zero-padded tiling (:478-560,
  ``patchify`` :22-42), the blank-patch test (:746-756), the labels, and the hot loop (:366-384) --
  per patch log-amplitude, phase, forward-difference gradient magnitude with per-patch min-max,
  fixed-range log-amp scaling, float32 cast and ImageNet normalisation -> NHWC float32.  The
  waterfall and its flags are uploaded once; no patch is ever materialised on the host.
* host: a table with one 16-byte entry per patch (plane, view, tile origin) -- its construction in
  the reference's patch order, the keep filter, the global-RNG shuffle (:758-763, so datasets are
  reproducible against the reference under ``np.random.seed``) and ``num_patches`` truncation.
Other inputs are tiled by the host (NumPy views) and processed per patch stack on the GPU: REAL
input goes through ``rfi_preprocess_real`` (median normalise :646-670, SQRT/LOG10 stretch with the
MAD of the finite values replacing infinities :672-706, MAD flags :708-745, channels :608-644), and
flag-less complex input gets its MAD flags from ``rfi_mad_flags``; the order statistics are exact
(radix selection on the ordered 64-bit image of the doubles).
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


def _patch_table(n_planes, C, T, rot, ps):
    """One (plane, view, row0, col0) entry per patch, in the reference's order: plane-major, then
    view (:413-446), then row-major tiles of the zero-padded view (:478-560)."""
    views = (0,) if rot < 2 else ((0, 1) if rot < 4 else (0, 1, 2, 3))
    whole = C <= ps and T <= ps          # small waterfalls are used whole, unpadded (:340-347)
    rows = []
    for plane in range(n_planes):
        for v in views:
            hv, wv = (T, C) if v >= 2 else (C, T)
            if whole:
                rows.append((plane, v, 0, 0))
                continue
            for r0 in range(0, max(hv, ps), ps):
                if r0 >= hv and r0 > 0:
                    break
                for c0 in range(0, max(wv, ps), ps):
                    if c0 >= wv and c0 > 0:
                        break
                    rows.append((plane, v, r0, c0))
    return np.asarray(rows, dtype=np.int32).reshape(-1, 4)


def select_patches(ctx, d_flags, n_planes, Cn, Tn, rot, ps, num_patches=None, inference_mode=False):
    """Patch table of a device-resident flag stack after the blank-patch test (GPU) and the
    global-RNG shuffle (host, the reference's ``np.random.permutation``, :758-763)."""
    table = _patch_table(n_planes, Cn, Tn, rot, ps)
    if not inference_mode:
        keep = np.zeros(len(table), dtype=np.uint8)
        if len(table):
            check(lib.rfi_patch_any_flag(ctx.handle, C.c_void_p(d_flags.ptr), DEVICE, n_planes, Cn, Tn,
                                         table.ctypes.data_as(C.c_void_p), len(table), ps,
                                         keep.ctypes.data_as(C.c_void_p)))
        keep = keep.astype(bool)
        if keep.any():
            table = table[keep]
        table = table[np.random.permutation(len(table))]
    if num_patches and num_patches < len(table):
        table = table[:num_patches]
    return np.ascontiguousarray(table)


def gather_patches(ctx, d_planes, d_flags, code, n_planes, Cn, Tn, table, ps, images, labels):
    """Run the gather kernels for `table`; `images` / `labels` are NumPy arrays (host results) or
    DeviceArrays (results stay in HBM); labels may be None."""
    from ..runtime import DeviceArray

    def ptr_mem(a):
        if a is None:
            return None, HOST
        if isinstance(a, DeviceArray):
            return C.c_void_p(a.ptr), DEVICE
        return a.ctypes.data_as(C.c_void_p), HOST
    ip, im = ptr_mem(images)
    lp, lm = ptr_mem(labels)
    if len(table):
        check(lib.rfi_preprocess_gather(ctx.handle, C.c_void_p(d_planes.ptr), DEVICE, code, n_planes, Cn, Tn,
                                        C.c_void_p(d_flags.ptr), DEVICE, table.ctypes.data_as(C.c_void_p), len(table),
                                        ps, ip, im, lp, lm))


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

    # ---- device path: views, tiling, blank-patch test, labels and channels straight from the waterfall
    def _create_on_device(self, patch_size, rot, num_patches, inference_mode, metadata):
        data = self.data
        B, P, Cn, Tn = data.shape
        whole = Cn <= patch_size and Tn <= patch_size
        if whole and Cn != Tn and rot >= 4:
            return None                       # ragged views of a small non-square waterfall: host path
        ps_h, ps_w = (Cn, Tn) if whole else (patch_size, patch_size)
        if ps_h != ps_w:
            return None
        ps = ps_h
        code = {np.dtype(np.complex128): C128, np.dtype(np.complex64): C64}.get(data.dtype)
        if code is None:
            data, code = data.astype(np.complex128), C128
        planes = np.ascontiguousarray(data.reshape(B * P, Cn, Tn))
        fl = np.asarray(self.flags)
        fl = fl[np.newaxis, ...] if fl.ndim == 3 else fl
        if fl.shape != self.data.shape:
            return None
        flags = np.ascontiguousarray(fl.reshape(B * P, Cn, Tn) != 0).view(np.uint8)
        ctx = Context.get(self._device)
        d_planes, d_flags = ctx.to_device(planes), ctx.to_device(flags)
        if not whole:
            self.original_shapes = [((Tn, Cn) if v >= 2 else (Cn, Tn)) for v in
                                    ((0,) if rot < 2 else ((0, 1) if rot < 4 else (0, 1, 2, 3)))] * (B * P)
        table = select_patches(ctx, d_flags, B * P, Cn, Tn, rot, ps, num_patches, inference_mode)
        n = len(table)
        images = np.empty((n, ps, ps, 3), dtype=np.float32)
        labels = np.zeros((n, ps, ps), dtype=np.uint8)
        gather_patches(ctx, d_planes, d_flags, code, B * P, Cn, Tn, table, ps, images,
                       None if inference_mode else labels)
        self._table, self._rot, self._ps = table, rot, ps
        self.patches = self.patch_flags = None      # not materialised; see materialise_patches()
        metadata["original_shapes"] = getattr(self, "original_shapes", None)
        self.dataset = TorchDataset(torch.from_numpy(images), torch.from_numpy(labels), metadata)
        return self.dataset

    def materialise_patches(self):
        """Host copies of the complex patches / flag patches of the last device-path dataset, in
        dataset order (the reference keeps them as ``self.patches`` / ``self.patch_flags``)."""
        if getattr(self, "_table", None) is None:
            return self.patches, self.patch_flags
        B, P, Cn, Tn = self.data.shape
        planes = self.data.reshape(B * P, Cn, Tn)
        fl = np.asarray(self.flags)
        fl = (fl[np.newaxis, ...] if fl.ndim == 3 else fl).reshape(B * P, Cn, Tn)
        ps = self._ps

        def cut(src, e):
            plane, v, r0, c0 = (int(x) for x in e)
            view = src[plane]
            view = view[::-1, :] if v == 1 else (view.T if v == 2 else (view.T[::-1, :] if v == 3 else view))
            out = np.zeros((ps, ps), dtype=src.dtype)
            blk = view[r0:r0 + ps, c0:c0 + ps]
            out[:blk.shape[0], :blk.shape[1]] = blk
            return out
        self.patches = np.array([cut(planes, e) for e in self._table])
        self.patch_flags = np.array([cut(fl, e) for e in self._table])
        return self.patches, self.patch_flags

    def create_dataset(self, patch_size=128, stretch=None, flag_sigma=5, use_custom_flags=True,
                       num_patches=None, normalize_before_stretch=True, normalize_after_stretch=False,
                       num_workers=4, enable_augmentation=True, augmentation_rotations=4,
                       inference_mode=False, on_device_tiling=True):
        del num_workers                       # the hot loop runs on the GPU, no worker pool
        rot = augmentation_rotations if (enable_augmentation and augmentation_rotations > 1) else 1
        have_flags = use_custom_flags and self.flags is not None
        self._table = None
        if on_device_tiling and have_flags and np.iscomplexobj(self.data):
            metadata = {"patch_size": patch_size, "stretch": stretch, "flag_sigma": flag_sigma,
                        "normalize_before_stretch": normalize_before_stretch,
                        "normalize_after_stretch": normalize_after_stretch,
                        "augmentation_rotations": augmentation_rotations}
            ds = self._create_on_device(patch_size, rot, num_patches, inference_mode, metadata)
            if ds is not None:
                return ds
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
        is_real = not np.iscomplexobj(patches)
        need_mad = pflags is None and not inference_mode
        images = None
        # float64 real input is processed in float64 on the device, float32 real input in float32 arithmetic (every result
        # rounded to float32: what NumPy does on a float32 array, preprocessor.py:608-706); other real dtypes: host
        if on_device_tiling and len(patches) and not (is_real and patches.dtype not in (np.float64, np.float32)):
            # order-statistic branches on the GPU: the real-input pipeline (median normalise, stretch,
            # MAD flags, channels) in one call; for flag-less complex input the MAD flags of |z|
            if is_real:
                images, mad = self._real_on_device(patches, stretch, normalize_before_stretch,
                                                   normalize_after_stretch, flag_sigma if need_mad else None)
                if need_mad:
                    pflags = mad
            elif need_mad:
                pflags = self._mad_flags_on_device(patches, flag_sigma)
        else:
            if is_real:
                patches = self._real_pipeline(patches, stretch, normalize_before_stretch, normalize_after_stretch)
            if need_mad:
                pflags = self._mad_flags(patches, flag_sigma)
        if inference_mode:
            pflags = np.zeros(patches.shape, dtype=np.uint8)
        sel = np.arange(len(patches))
        if not inference_mode:
            keep = pflags.reshape(len(pflags), -1).any(axis=1)
            if keep.any():
                sel = sel[keep]
            sel = sel[np.random.permutation(len(sel))]         # global RNG, as the reference (:760)
        if num_patches and num_patches < len(sel):
            sel = sel[:num_patches]
        pflags = pflags[sel]
        if images is not None:                                  # channels already computed for every patch
            images = np.ascontiguousarray(images[sel])
            self.patches, self.patch_flags = None, pflags       # processed real patches stay on the device
        else:
            patches = patches[sel]
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

    # ---- order-statistic branches on the GPU (librfi_hip.so rfi_preprocess_real / rfi_mad_flags)
    def _real_on_device(self, patches, stretch, before, after, sigma):
        n, ph, pw = patches.shape
        code = F32 if patches.dtype == np.float32 else F64
        patches = np.ascontiguousarray(patches, dtype=np.float32 if code == F32 else np.float64)
        images = np.empty((n, ph, pw, 3), dtype=np.float32)
        flags = np.empty((n, ph, pw), dtype=np.uint8) if sigma is not None else None
        ctx = Context.get(self._device)
        check(lib.rfi_preprocess_real(ctx.handle, patches.ctypes.data_as(C.c_void_p), HOST, code, n, ph, pw,
                                      {None: 0, "": 0, "SQRT": 1, "LOG10": 2}[stretch], 1 if before else 0,
                                      1 if after else 0, float(sigma if sigma is not None else 0.0),
                                      images.ctypes.data_as(C.c_void_p), HOST,
                                      flags.ctypes.data_as(C.c_void_p) if flags is not None else None, HOST))
        return images, (flags.astype(bool) if flags is not None else None)

    def _mad_flags_on_device(self, patches, sigma):
        n, ph, pw = patches.shape
        code = {np.dtype(np.complex128): C128, np.dtype(np.complex64): C64, np.dtype(np.float64): F64,
                np.dtype(np.float32): F32}.get(patches.dtype)
        if code is None:
            patches, code = patches.astype(np.complex128), C128
        patches = np.ascontiguousarray(patches)
        flags = np.empty((n, ph, pw), dtype=np.uint8)
        ctx = Context.get(self._device)
        check(lib.rfi_mad_flags(ctx.handle, patches.ctypes.data_as(C.c_void_p), HOST, code, n, ph, pw, float(sigma),
                                flags.ctypes.data_as(C.c_void_p), HOST))
        return flags.astype(bool)

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
