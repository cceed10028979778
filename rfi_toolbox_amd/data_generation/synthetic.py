"""Synthetic complex waterfalls with exact RFI masks.

Same physical model and parameter distributions as the reference generator
(rfi_toolbox/data_generation/synthetic_generator.py): noise N(1, 0.1) mJy (:553), optional t^8
bandpass roll-off on the outer 10 % of channels (:658-673), RFI amplitude U(1000, 10000)*1000
(:578), the six injector families (:675-815), polarisation 1 correlated at 0.8, polarisations >= 2
noise only (:626-644), uniform random phase (:647-648).  It is NOT the reference's RNG stream: it
draws from a private ``numpy.random.Generator`` (the reference uses the global legacy RNG), so
samples are distribution-equivalent, not bit-equal.

Two back ends share the event draw (a few dozen rectangles / sweeps per sample, host side):
``sample()`` rasterises and adds noise with NumPy on the host; ``sample_device(n)`` sends only the
event table and lets librfi_hip.so (``rfi_generate_waterfalls``, csrc/synth.hip) write the complex
waterfalls and masks straight into HBM with a counter-based per-pixel random stream, where
``Preprocessor``'s gather kernels and the training step consume them without a host round trip.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

DEFAULT_COUNTS = {"narrowband_persistent": 1, "broadband_persistent": 1, "frequency_sweep": 1}
# synthetic_train_4k.yaml:15-20 densities are for 1024x1024; scale with area for small waterfalls
YAML_4K_COUNTS = {"narrowband_persistent": 20, "broadband_persistent": 5, "frequency_sweep": 1,
                  "narrowband_bursty": 20, "broadband_bursty": 5}


class SyntheticWaterfalls:
    def __init__(self, num_channels=128, num_times=128, num_polarizations=1, counts=None, noise_mjy=1.0,
                 rfi_power=(1000.0, 10000.0), bandpass=True, bandpass_order=8, pol_corr=0.8, seed=0):
        self.nc, self.nt, self.npol = num_channels, num_times, num_polarizations
        self.counts = dict(counts or DEFAULT_COUNTS)
        self.noise, self.power = noise_mjy, rfi_power
        self.bandpass, self.order, self.corr = bandpass, bandpass_order, pol_corr
        self.rng = np.random.default_rng(seed)

    # ---- injectors: each returns primitives (kind, r0, r1, c0, c1); kind 0 = channels [r0,r1) x times
    # [c0,c1); kind 1 = sweep from channel r0 to r1, width c0, power-law order c1 (the C ABI's rfi_event)
    def _primitives(self, kind):
        r, nc, nt = self.rng, self.nc, self.nt
        lo, hi = int(nc * 0.1), max(int(nc * 0.9), int(nc * 0.1) + 1)
        if kind == "narrowband_persistent":
            c, bw = r.integers(lo, hi), r.integers(1, 10)
            return [(0, max(0, c - bw // 2), min(nc, c + bw // 2 + 1), 0, nt)]
        if kind == "broadband_persistent":
            t = r.integers(int(nt * 0.1), max(int(nt * 0.9), int(nt * 0.1) + 1))
            tw = r.integers(5, 50)
            return [(0, 0, nc, max(0, t - tw // 2), min(nt, t + tw // 2))]
        if kind == "narrowband_intermittent":
            c, bw = r.integers(lo, hi), r.integers(2, 15)
            period, duty = r.integers(20, 200), r.uniform(0.1, 0.5)
            f0, f1 = max(0, c - bw // 2), min(nc, c + bw // 2)
            return [(0, f0, f1, t, min(nt, t + int(period * duty))) for t in range(0, nt, period)]
        if kind == "narrowband_bursty":
            c, bw, nb = r.integers(lo, hi), r.integers(2, 20), min(r.integers(3, 15), nt)
            f0, f1 = max(0, c - bw // 2), min(nc, c + bw // 2)
            ts, ws = r.choice(nt, nb, replace=False), r.integers(2, 20, nb)
            return [(0, f0, f1, max(0, t - w // 2), min(nt, t + w // 2)) for t, w in zip(ts, ws)]
        if kind == "broadband_bursty":
            nb = min(r.integers(2, 10), nt)
            ts, ws = r.choice(nt, nb, replace=False), r.integers(1, 5, nb)
            return [(0, 0, nc, max(0, t - w // 2), min(nt, t + w // 2)) for t, w in zip(ts, ws)]
        if kind == "frequency_sweep":
            f0 = r.integers(lo, max(int(nc * 0.5), lo + 1))
            f1 = r.integers(int(nc * 0.5), max(int(nc * 0.9), int(nc * 0.5) + 1))
            bw, order = r.integers(2, 10), r.choice([1, 2])
            return [(1, int(f0), int(f1), int(bw), int(order))]
        raise ValueError(f"unknown RFI type {kind}")

    def _rects(self, kind):
        """The primitives of one event as (channel slice, time slice) rectangles (host rasterisation)."""
        nc, nt, out = self.nc, self.nt, []
        for k, r0, r1, c0, c1 in self._primitives(kind):
            if k == 0:
                out.append((slice(int(r0), int(r1)), slice(int(c0), int(c1))))
                continue
            for t in range(nt):
                x = t / nt
                c = int(r0 + (r1 - r0) * (x * x if c1 == 2 else x))
                out.append((slice(max(0, c - c0 // 2), min(nc, c + c0 // 2)), slice(t, t + 1)))
        return out

    def draw_events(self):
        """One sample's event table: [(kind, r0, r1, c0, c1, amp)], amplitude U(lo, hi) * 1000 per event."""
        ev = []
        for kind, cnt in self.counts.items():
            if isinstance(cnt, (list, tuple)):
                cnt = self.rng.integers(cnt[0], cnt[1] + 1)
            for _ in range(int(cnt)):
                amp = float(self.rng.uniform(*self.power) * 1000.0)
                ev += [(int(k), int(a), int(b), int(c), int(d), amp) for k, a, b, c, d in self._primitives(kind)]
        return ev

    def sample_device(self, n_samples, device=None, dtype=np.complex128, seed=None):
        """n_samples waterfalls generated IN HBM -> (planes DeviceArray (n, npol, nc, nt) complex,
        flags DeviceArray uint8 same shape, events per sample).  Only the event table crosses PCIe."""
        from .._lib import C128, C64, DEVICE, HOST, check, lib
        from ..runtime import Context
        ctx = Context.get(device)
        events = [self.draw_events() for _ in range(n_samples)]
        rec = np.dtype([("kind", "<i4"), ("r0", "<i4"), ("r1", "<i4"), ("c0", "<i4"), ("c1", "<i4"), ("pad", "<i4"),
                        ("amp", "<f8")])
        flat = np.zeros(max(1, sum(len(e) for e in events)), dtype=rec)
        offs = np.zeros(n_samples + 1, dtype=np.int32)
        k = 0
        for i, ev in enumerate(events):
            for (kind, r0, r1, c0, c1, amp) in ev:
                flat[k] = (kind, r0, r1, c0, c1, 0, amp)
                k += 1
            offs[i + 1] = k
        code = {np.dtype(np.complex128): C128, np.dtype(np.complex64): C64}[np.dtype(dtype)]
        shape = (n_samples, self.npol, self.nc, self.nt)
        planes, flags = ctx.empty(shape, dtype), ctx.empty(shape, np.uint8)
        if seed is None:
            seed = int(self.rng.integers(0, 2 ** 63))
        check(lib.rfi_generate_waterfalls(ctx.handle, seed, n_samples, self.npol, self.nc, self.nt, float(self.noise),
                                          1 if self.bandpass else 0, int(self.order), float(self.corr),
                                          flat.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), code,
                                          C.c_void_p(planes.ptr), DEVICE, C.c_void_p(flags.ptr), DEVICE))
        return planes, flags, events

    def sample(self):
        """-> waterfall (1, npol, nc, nt) complex128, mask (1, npol, nc, nt) bool"""
        r, nc, nt = self.rng, self.nc, self.nt
        base = r.normal(self.noise, self.noise * 0.1, (nc, nt))
        if self.bandpass:
            bp = np.ones(nc)
            edge = int(nc * 0.1)
            if edge:
                t = (np.arange(edge) / edge) ** self.order
                bp[:edge] = t
                bp[nc - edge:] = t[::-1]
            base = base * bp[:, None]
        sig = np.zeros((nc, nt))
        mask = np.zeros((nc, nt), dtype=bool)
        for kind, cnt in self.counts.items():
            if isinstance(cnt, (list, tuple)):
                cnt = r.integers(cnt[0], cnt[1] + 1)
            for _ in range(int(cnt)):
                amp = r.uniform(*self.power) * 1000.0
                one = np.zeros((nc, nt))
                for fs, ts in self._rects(kind):
                    one[fs, ts] = amp
                    mask[fs, ts] = True
                sig += one
        pols, masks = [], []
        for p in range(self.npol):
            if p == 0:
                real, m = base + sig, mask
            elif p == 1:
                real = self.corr * sig + (1 - self.corr) * r.normal(0, self.noise * 0.1, sig.shape) + base
                m = mask
            else:
                real, m = r.normal(self.noise, self.noise * 0.1, (nc, nt)), np.zeros_like(mask)
            pols.append(real * np.exp(1j * r.uniform(0, 2 * np.pi, real.shape)))
            masks.append(m.copy())
        return np.stack(pols)[None], np.stack(masks)[None]


def make_training_patches(n_patches, size=128, seed=0, device=None, counts=None):
    """n_patches NHWC float32 images + uint8 labels of size x size, produced like the reference
    pipeline: synthetic waterfall -> Preprocessor (4 views, custom exact flags) on the GPU."""
    from ..preprocessing import Preprocessor
    if counts is None:
        scale = (size * size) / (1024.0 * 1024.0)
        counts = {k: max(1, int(round(v * scale * 4))) for k, v in YAML_4K_COUNTS.items()}
    gen = SyntheticWaterfalls(size, size, 1, counts=counts, seed=seed)
    imgs, labs, state = [], [], np.random.get_state()
    np.random.seed(seed)                       # Preprocessor shuffles with the global RNG, as the reference
    try:
        have = 0
        while have < n_patches:
            w, m = gen.sample()
            ds = Preprocessor(w, flags=m, device=device).create_dataset(patch_size=size, num_workers=0)
            imgs.append(ds.images.numpy())
            labs.append(ds.labels.numpy())
            have += len(ds)
    finally:
        np.random.set_state(state)
    return np.concatenate(imgs)[:n_patches], np.concatenate(labs)[:n_patches]


def make_training_patches_device(n_patches, size=128, seed=0, device=None, counts=None, dtype=np.complex64):
    """The whole input pipeline in HBM: waterfalls generated on the GPU (``sample_device``), views /
    tiling / blank-patch test / channels / labels by the gather kernels, results left on the device.
    -> (images DeviceArray (n, size, size, 3) float32, labels DeviceArray (n, size, size) uint8).
    Only event tables and 16-byte patch-table entries cross PCIe."""
    from .._lib import C128, C64
    from ..preprocessing.preprocessor import gather_patches, select_patches
    from ..runtime import Context
    ctx = Context.get(device)
    if counts is None:
        scale = (size * size) / (1024.0 * 1024.0)
        counts = {k: max(1, int(round(v * scale * 4))) for k, v in YAML_4K_COUNTS.items()}
    gen = SyntheticWaterfalls(size, size, 1, counts=counts, seed=seed)
    code = {np.dtype(np.complex128): C128, np.dtype(np.complex64): C64}[np.dtype(dtype)]
    images, labels = ctx.empty((n_patches, size, size, 3), np.float32), ctx.empty((n_patches, size, size), np.uint8)
    state = np.random.get_state()
    np.random.seed(seed)
    try:
        have = 0
        while have < n_patches:
            n_wf = max(1, (n_patches - have + 3) // 4)
            planes, flags, _ = gen.sample_device(n_wf, device=device, dtype=dtype)
            table = select_patches(ctx, flags, n_wf, size, size, 4, size)
            table = table[: n_patches - have]
            if len(table) == 0:
                continue
            from ..runtime import DeviceArray

            class _View(DeviceArray):          # window into the output arrays (no ownership)
                def __init__(self, ptr):
                    self.ptr = ptr

                def __del__(self):
                    pass
            gather_patches(ctx, planes, flags, code, n_wf, size, size, np.ascontiguousarray(table), size,
                           _View(images.ptr + have * size * size * 3 * 4), _View(labels.ptr + have * size * size))
            have += len(table)
    finally:
        np.random.set_state(state)
    ctx.synchronize()
    return images, labels
