"""Thin host runtime over the C ABI: one Context per GPU (one HIP stream), device buffers,
and array plumbing between NumPy / torch and raw pointers.  No compute happens in Python."""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

from . import _lib
from ._lib import DEVICE, HOST, check, lib

try:                                    # torch is plumbing only (tensor I/O, init RNG, distributed)
    import torch
except Exception:                       # pragma: no cover
    torch = None


def _parse_device(device) -> int:
    if device is None:
        return int(os.environ.get("LOCAL_RANK", "0"))
    if isinstance(device, int):
        return device
    s = str(device)
    if s == "cpu":
        raise RuntimeError("rfi_toolbox_amd runs on MI355X GPUs only; there is no CPU path "
                           "(use the reference rfi_toolbox for CPU execution)")
    if s in ("cuda", "hip", "gpu"):
        return int(os.environ.get("LOCAL_RANK", "0"))
    if ":" in s:
        return int(s.split(":")[1])
    raise ValueError(f"unknown device {device!r}")


class Context:
    """Owns a rfi_ctx (GPU + HIP stream).  Not thread-safe: one host thread per context."""

    _cache: dict = {}
    _lock = threading.Lock()

    def __init__(self, device=None):
        self.device_index = _parse_device(device)
        h = C.c_void_p()
        check(lib.rfi_ctx_create(self.device_index, C.byref(h)))
        self.handle = h
        self._comm = False

    @classmethod
    def get(cls, device=None) -> "Context":
        idx = _parse_device(device)
        with cls._lock:
            ctx = cls._cache.get((os.getpid(), idx))
            if ctx is None:
                ctx = cls(idx)
                cls._cache[(os.getpid(), idx)] = ctx
            return ctx

    # ---- memory
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(lib.rfi_malloc(self.handle, max(int(nbytes), 16), C.byref(p)))
        return p.value

    def free(self, ptr: int):
        check(lib.rfi_free(self.handle, C.c_void_p(ptr)))

    def synchronize(self):
        check(lib.rfi_ctx_synchronize(self.handle))

    def set_overlap(self, on: bool):
        """Backward-pass overlap of the weight-gradient kernels (side stream); default on."""
        check(lib.rfi_ctx_set_overlap(self.handle, 1 if on else 0))

    def stream_ptr(self) -> int:
        s = C.c_void_p()
        check(lib.rfi_ctx_stream(self.handle, C.byref(s)))
        return s.value or 0

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        check(lib.rfi_ctx_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def to_device(self, arr: np.ndarray) -> "DeviceArray":
        arr = np.ascontiguousarray(arr)
        d = DeviceArray(self, arr.shape, arr.dtype)
        check(lib.rfi_memcpy(self.handle, C.c_void_p(d.ptr), DEVICE, arr.ctypes.data_as(C.c_void_p), HOST,
                             arr.nbytes))
        return d

    def empty(self, shape, dtype=np.float32) -> "DeviceArray":
        return DeviceArray(self, shape, np.dtype(dtype))

    # ---- timing / profile
    def timer_start(self):
        check(lib.rfi_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(lib.rfi_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def profile(self, on: bool):
        check(lib.rfi_profile_enable(self.handle, 1 if on else 0))

    def profile_reset(self):
        check(lib.rfi_profile_reset(self.handle))

    def profile_dump(self, path: str):
        check(lib.rfi_profile_dump(self.handle, path.encode()))

    def profile_report(self) -> dict:
        out = {}
        for f in range(lib.rfi_profile_family_count()):
            n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            check(lib.rfi_profile_get(self.handle, f, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
            if n.value:
                out[lib.rfi_profile_family_name(f).decode()] = {
                    "launches": n.value, "ms": ms.value, "flops": fl.value, "bytes": by.value}
        return out

    # ---- RCCL
    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = C.create_string_buffer(unique_id, 128)
        check(lib.rfi_comm_init(self.handle, buf, rank, world))
        self._comm = True

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(lib.rfi_comm_unique_id(buf))
        return buf.raw

    def comm_emulate(self, world: int):
        """Single-GPU test mode of the bucketed gradient exchange (0 = off): see include/rfi_hip.h."""
        check(lib.rfi_comm_emulate(self.handle, int(world)))

    def comm_destroy(self):
        if self._comm:
            check(lib.rfi_comm_destroy(self.handle))
            self._comm = False


class DeviceArray:
    """A typed HBM buffer owned by a Context (freed on garbage collection)."""

    def __init__(self, ctx: Context, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.ptr = ctx.malloc(self.nbytes)

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            check(lib.rfi_memcpy(self.ctx.handle, out.ctypes.data_as(C.c_void_p), HOST, C.c_void_p(self.ptr),
                                 DEVICE, self.nbytes))
        return out

    def copy_from(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes == self.nbytes
        check(lib.rfi_memcpy(self.ctx.handle, C.c_void_p(self.ptr), DEVICE, arr.ctypes.data_as(C.c_void_p),
                             HOST, self.nbytes))

    def zero_(self):
        check(lib.rfi_memset(self.ctx.handle, C.c_void_p(self.ptr), 0, self.nbytes))

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                self.ctx.free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def as_pointer(x, dtype, ctx: Context):
    """-> (ptr, mem, keepalive).  Accepts DeviceArray, NumPy array, torch CPU / CUDA tensor."""
    if isinstance(x, DeviceArray):
        if x.dtype != np.dtype(dtype):
            raise TypeError(f"device array has dtype {x.dtype}, expected {np.dtype(dtype)}")
        return x.ptr, DEVICE, x
    if is_torch(x):
        tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.uint8): torch.uint8}[np.dtype(dtype)]
        if x.is_cuda:
            if x.device.index not in (None, ctx.device_index):
                raise RuntimeError(f"tensor is on {x.device}, context on GPU {ctx.device_index}")
            t = x.detach().to(tdt).contiguous()
            torch.cuda.current_stream(t.device).synchronize()     # hand over to the ctx stream
            return t.data_ptr(), DEVICE, t
        a = np.ascontiguousarray(x.detach().to(tdt).numpy())
        return a.ctypes.data, HOST, a
    a = np.ascontiguousarray(np.asarray(x), dtype=dtype)
    return a.ctypes.data, HOST, a
