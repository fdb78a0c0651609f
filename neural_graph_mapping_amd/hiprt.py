"""Minimal torch-free HIP runtime access (ctypes on libamdhip64) for the C-ABI harness:
device buffers backed by numpy on the host side, streams, events.  Used by the fast GPU checks
and by bench.py's kernel timers (HIP events on the launch stream)."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("libamdhip64.so not found")
        _hip.hipGetErrorString.restype = C.c_char_p
        _hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        _hip.hipEventSynchronize.argtypes = [C.c_void_p]
        _hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        _hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        _hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    return _hip


def _chk(err, what):
    if err != 0:
        raise RuntimeError(f"{what}: HIP error {err}: {hip().hipGetErrorString(err).decode()}")


def device_count():
    n = C.c_int(0)
    try:
        err = hip().hipGetDeviceCount(C.byref(n))
    except RuntimeError:
        return 0
    return n.value if err == 0 else 0


def set_device(i):
    _chk(hip().hipSetDevice(int(i)), "hipSetDevice")


def synchronize():
    _chk(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")


class DeviceArray:
    """A device allocation with numpy shape/dtype metadata."""

    def __init__(self, shape, dtype=np.float32, zero=True):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        _chk(hip().hipMalloc(C.byref(p), max(self.nbytes, 16)), "hipMalloc")
        self.ptr = p.value
        if zero:
            _chk(hip().hipMemset(self.ptr, 0, max(self.nbytes, 16)), "hipMemset")

    @staticmethod
    def from_numpy(a):
        a = np.ascontiguousarray(a)
        d = DeviceArray(a.shape, a.dtype, zero=False)
        if a.nbytes:
            _chk(hip().hipMemcpy(d.ptr, a.ctypes.data, a.nbytes, 1), "hipMemcpy H2D")
        return d

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        synchronize()
        if self.nbytes:
            _chk(hip().hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, 2), "hipMemcpy D2H")
        return out

    def fill_bytes(self, value=0):
        _chk(hip().hipMemset(self.ptr, value, max(self.nbytes, 16)), "hipMemset")

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def to_dev(a, dtype=None):
    if a is None:
        return None
    a = np.asarray(a)
    if dtype is not None:
        a = a.astype(dtype)
    return DeviceArray.from_numpy(a)


def ptr(d):
    return None if d is None else d.ptr


class Stream:
    def __init__(self):
        p = C.c_void_p()
        _chk(hip().hipStreamCreate(C.byref(p)), "hipStreamCreate")
        self.handle = p.value

    def synchronize(self):
        _chk(hip().hipStreamSynchronize(self.handle), "hipStreamSynchronize")


class Event:
    def __init__(self):
        p = C.c_void_p()
        _chk(hip().hipEventCreate(C.byref(p)), "hipEventCreate")
        self.handle = p.value

    def record(self, stream=None):
        _chk(hip().hipEventRecord(self.handle, stream), "hipEventRecord")

    def synchronize(self):
        _chk(hip().hipEventSynchronize(self.handle), "hipEventSynchronize")

    def elapsed_ms(self, end):
        ms = C.c_float(0)
        _chk(hip().hipEventElapsedTime(C.byref(ms), self.handle, end.handle), "hipEventElapsedTime")
        return ms.value
