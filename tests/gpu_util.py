"""Helpers shared by the -m gpu parity tests (call the HIP kernels through the C ABI)."""
import ctypes as C

import numpy as np
import torch

from rfi_toolbox_amd._lib import IMPL_AUTO, IMPL_DIRECT, IMPL_MFMA, check, lib  # noqa: F401
from rfi_toolbox_amd.runtime import Context


def ctx():
    return Context.get(0)


def P(d):
    return C.c_void_p(d.ptr) if d is not None else None


def nhwc(t):            # torch NCHW -> numpy NHWC
    return np.ascontiguousarray(t.permute(0, 2, 3, 1).numpy())


def nchw(a):            # numpy NHWC -> torch NCHW
    return torch.from_numpy(np.ascontiguousarray(a)).permute(0, 3, 1, 2).contiguous()


def rel_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))
