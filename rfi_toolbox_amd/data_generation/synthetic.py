"""Synthetic complex waterfalls with exact RFI masks.

Same physical model and parameter distributions as the reference generator
(rfi_toolbox/data_generation/synthetic_generator.py): noise N(1, 0.1) mJy (:553), optional t^8
bandpass roll-off on the outer 10 % of channels (:658-673), RFI amplitude U(1000, 10000)*1000
(:578), the six injector families (:675-815), polarisation 1 correlated at 0.8, polarisations >= 2
noise only (:626-644), uniform random phase (:647-648).  It is NOT the reference's RNG stream: it
draws from a private ``numpy.random.Generator`` (the reference uses the global legacy RNG), so
samples are distribution-equivalent, not bit-equal.  Host-side NumPy; used to feed benchmarks.
"""
from __future__ import annotations

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

    # ---- injectors: each returns (freq slice/rows, time slice/cols) rectangles to fill
    def _rects(self, kind):
        r, nc, nt = self.rng, self.nc, self.nt
        lo, hi = int(nc * 0.1), max(int(nc * 0.9), int(nc * 0.1) + 1)
        if kind == "narrowband_persistent":
            c, bw = r.integers(lo, hi), r.integers(1, 10)
            return [(slice(max(0, c - bw // 2), min(nc, c + bw // 2 + 1)), slice(0, nt))]
        if kind == "broadband_persistent":
            t = r.integers(int(nt * 0.1), max(int(nt * 0.9), int(nt * 0.1) + 1))
            tw = r.integers(5, 50)
            return [(slice(0, nc), slice(max(0, t - tw // 2), min(nt, t + tw // 2)))]
        if kind == "narrowband_intermittent":
            c, bw = r.integers(lo, hi), r.integers(2, 15)
            period, duty = r.integers(20, 200), r.uniform(0.1, 0.5)
            fs = slice(max(0, c - bw // 2), min(nc, c + bw // 2))
            return [(fs, slice(t, min(nt, t + int(period * duty)))) for t in range(0, nt, period)]
        if kind == "narrowband_bursty":
            c, bw, nb = r.integers(lo, hi), r.integers(2, 20), min(r.integers(3, 15), nt)
            fs = slice(max(0, c - bw // 2), min(nc, c + bw // 2))
            ts, ws = r.choice(nt, nb, replace=False), r.integers(2, 20, nb)
            return [(fs, slice(max(0, t - w // 2), min(nt, t + w // 2))) for t, w in zip(ts, ws)]
        if kind == "broadband_bursty":
            nb = min(r.integers(2, 10), nt)
            ts, ws = r.choice(nt, nb, replace=False), r.integers(1, 5, nb)
            return [(slice(0, nc), slice(max(0, t - w // 2), min(nt, t + w // 2))) for t, w in zip(ts, ws)]
        if kind == "frequency_sweep":
            f0 = r.integers(lo, max(int(nc * 0.5), lo + 1))
            f1 = r.integers(int(nc * 0.5), max(int(nc * 0.9), int(nc * 0.5) + 1))
            bw, order = r.integers(2, 10), r.choice([1, 2])
            out = []
            for t in range(nt):
                c = int(f0 + (f1 - f0) * (t / nt) ** order)
                out.append((slice(max(0, c - bw // 2), min(nc, c + bw // 2)), slice(t, t + 1)))
            return out
        raise ValueError(f"unknown RFI type {kind}")

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
