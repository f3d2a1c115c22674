"""Minimal ctypes view of the HIP runtime libalfa_hot is linked with: device buffers for tests that hand DEVICE pointers
to the C ABI (test infrastructure; torch ships a second copy of the runtime, and two runtimes in one process do not mix)."""
import ctypes as C

import numpy as np

_rt = None


def rt():
    global _rt
    if _rt is None:
        from alfalfa_amd import ffi
        ffi.lib()                                   # maps libamdhip64.so.7 from /opt/rocm
        _rt = C.CDLL("libamdhip64.so.7")
        _rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _rt.hipFree.argtypes = [C.c_void_p]
        _rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    return _rt


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: hipError %d" % (what, rc))


class DevArray:
    """a device buffer holding a copy of a numpy array"""

    def __init__(self, arr=None, nbytes=None, fill=None):
        r = rt()
        self.nbytes = int(arr.nbytes if arr is not None else nbytes)
        p = C.c_void_p()
        _ck(r.hipMalloc(C.byref(p), max(self.nbytes, 16)), "hipMalloc")
        self.ptr = p.value
        if arr is not None:
            a = np.ascontiguousarray(arr)
            _ck(r.hipMemcpy(self.ptr, a.ctypes.data, self.nbytes, 1), "hipMemcpy H2D")
        elif fill is not None:
            _ck(r.hipMemset(self.ptr, fill, self.nbytes), "hipMemset")

    def get(self, dtype, shape=None):
        out = np.zeros(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _ck(rt().hipDeviceSynchronize(), "hipDeviceSynchronize")
        _ck(rt().hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, 2), "hipMemcpy D2H")
        return out.reshape(shape) if shape else out

    def free(self):
        if self.ptr:
            rt().hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sync():
    _ck(rt().hipDeviceSynchronize(), "hipDeviceSynchronize")
