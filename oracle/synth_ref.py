"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the on-device synthetic generator
(rfi_toolbox_amd/csrc/synth.hip).  Only tests/ may import this.

The physical model is the reference's ``SyntheticDataGenerator._generate_single_sample``
(rfi_toolbox/data_generation/synthetic_generator.py:520-815): noise N(n, 0.1 n) (:553), t^order
bandpass on the outer 10 % of channels (:658-673), constant-amplitude events summed into the signal
with their union as the exact mask (:675-815), polarisation mixing (:626-644), uniform random phase
(:647-648).  The random STREAM is the build's own (Philox4x32-10 counted by pixel coordinates; the
reference's is NumPy's sequential global generator and cannot be matched by a parallel device
kernel), so this oracle pins the device kernel bit-for-bit on mask / signal and to rounding on the
noise, while parity with the reference itself is distribution-level (tests/test_gpu_synth.py
compares moments against fixtures captured from the reference).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al., SC'11); counters are uint64 arrays holding 32-bit values."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) & MASK for v in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def _normal(a, b):
    r = np.sqrt(-2.0 * np.log((a.astype(np.float64) + 1.0) / 4294967296.0))
    return r * np.cos(6.283185307179586 * (b.astype(np.float64) / 4294967296.0))


def bandpass(C, order):
    edge = int(C * 0.1)
    g = np.ones(C)
    for i in range(edge):
        g[i] = (i / edge) ** order
        g[C - 1 - i] = (i / edge) ** order
    return g


def rasterise(events, C, T):
    """(signal float64 (C,T), mask bool (C,T)) of one sample's event list [(kind,r0,r1,c0,c1,amp)]."""
    sig = np.zeros((C, T))
    mask = np.zeros((C, T), dtype=bool)
    for kind, r0, r1, c0, c1, amp in events:
        if kind == 0:
            sig[r0:r1, c0:c1] += amp
            mask[r0:r1, c0:c1] = True
        else:
            for t in range(T):
                x = t / T
                c = int(r0 + (r1 - r0) * (x * x if c1 == 2 else x))
                lo, hi = max(0, c - c0 // 2), min(C, c + c0 // 2)
                sig[lo:hi, t] += amp
                mask[lo:hi, t] = True
    return sig, mask


def generate(seed, events_per_sample, n_pol, C, T, noise=1.0, use_bandpass=True, order=8, corr=0.8):
    """-> planes complex128 (n, n_pol, C, T), flags uint8 (n, n_pol, C, T)."""
    n = len(events_per_sample)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    per = C * T
    planes = np.zeros((n, n_pol, C, T), dtype=np.complex128)
    flags = np.zeros((n, n_pol, C, T), dtype=np.uint8)
    g = bandpass(C, order)[:, None] if use_bandpass else 1.0
    for s in range(n):
        spix = np.uint64(s) * np.uint64(per) + np.arange(per, dtype=np.uint64)
        lo, hi = spix & MASK, spix >> np.uint64(32)
        a = philox4x32_10(lo, hi, 0, 0, k0, k1)
        base = ((noise + 0.1 * noise * _normal(a[0], a[1])).reshape(C, T)) * g
        sig, mask = rasterise(events_per_sample[s], C, T)
        for p in range(n_pol):
            b = philox4x32_10(lo, hi, 1 + p, 0, k0, k1)
            npn = _normal(b[0], b[1]).reshape(C, T)
            if p == 0:
                real = base + sig
            elif p == 1:
                real = corr * sig + (1.0 - corr) * (0.1 * noise * npn) + base
            else:
                real = noise + 0.1 * noise * npn
            ph = 6.283185307179586 * (b[2].astype(np.float64) / 4294967296.0).reshape(C, T)
            planes[s, p] = real * np.cos(ph) + 1j * (real * np.sin(ph))
            if p < 2:
                flags[s, p] = mask
    return planes, flags
