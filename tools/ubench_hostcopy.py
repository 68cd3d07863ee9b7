"""Host <-> device copy rates that decide the SA_MEM_HOST path (csrc/sunode_amd.cpp): pageable vs pinned vs
hipHostRegister-on-the-fly, at the sizes of BASELINE config 2 (y_out 52 MB, stats 8 MB, y0 1 MB).
    python tools/ubench_hostcopy.py        (GPU box)"""
import ctypes
import time

import numpy as np

hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
hip.hipHostFree.argtypes = [ctypes.c_void_p]
D2H, H2D = 2, 1


def t(fn, reps=5):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); hip.hipDeviceSynchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3


for mb in (1, 8, 52):
    n = mb << 20
    d = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(d), n) == 0
    pageable = np.zeros(n, np.uint8)
    p = ctypes.c_void_p(); assert hip.hipHostMalloc(ctypes.byref(p), n, 0) == 0
    pinned = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(n,))
    pinned[:] = 1
    out = {}
    out["pageable D2H"] = t(lambda: hip.hipMemcpy(pageable.ctypes.data, d, n, D2H))
    out["pageable H2D"] = t(lambda: hip.hipMemcpy(d, pageable.ctypes.data, n, H2D))
    out["pinned D2H"] = t(lambda: hip.hipMemcpy(p, d, n, D2H))
    out["pinned H2D"] = t(lambda: hip.hipMemcpy(d, p, n, H2D))
    out["pinned D2H + memcpy to pageable"] = t(lambda: (hip.hipMemcpy(p, d, n, D2H), np.copyto(pageable, pinned)))

    def reg():
        assert hip.hipHostRegister(pageable.ctypes.data, n, 0) == 0
        hip.hipMemcpy(pageable.ctypes.data, d, n, D2H)
        hip.hipHostUnregister(pageable.ctypes.data)
    out["register + D2H + unregister"] = t(reg)
    out["np.zeros (fresh) + touch"] = t(lambda: np.zeros(n, np.uint8).__setitem__(slice(None, None, 4096), 1))
    def fresh():
        a = np.empty(n, np.uint8)           # (held until the copy is done)
        hip.hipMemcpy(a.ctypes.data, d, n, D2H)
        return a
    out["pageable D2H into a FRESH array"] = t(fresh)
    print("%3d MB: " % mb + "; ".join("%s %.2f ms" % kv for kv in out.items()))
    hip.hipHostFree(p)
