#!/usr/bin/env python3
"""Cold-start probe: what the driver charges for fresh device memory in a new process -- ONE hipMalloc
of S GB against S hipMallocs of 1 GB, first touch (memset) and second touch, free and re-allocate. Pure HIP through ctypes."""
import ctypes
import json
import sys
import time

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
GB = 1 << 30
S = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mode = sys.argv[2] if len(sys.argv) > 2 else "one"


def t(f):
    t0 = time.perf_counter()
    f()
    hip.hipDeviceSynchronize()
    return time.perf_counter() - t0


out = {"GB": S, "mode": mode}
out["init_s"] = t(lambda: hip.hipFree(None))
ptrs = []


def alloc():
    if mode == "one":
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), S * GB) == 0
        ptrs.append((p, S * GB))
    else:
        for _ in range(S):
            p = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(p), GB) == 0
            ptrs.append((p, GB))


def touch():
    for p, b in ptrs:
        hip.hipMemset(p, 0, b)


def free():
    for p, _ in ptrs:
        hip.hipFree(p)
    ptrs.clear()


out["alloc_s"] = t(alloc)
out["first_touch_s"] = t(touch)
out["second_touch_s"] = t(touch)
out["free_s"] = t(free)
out["realloc_s"] = t(alloc)
out["retouch_s"] = t(touch)
print(json.dumps(out))
