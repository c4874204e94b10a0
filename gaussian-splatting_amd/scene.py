"""Host-side scene loading through libgsx's host entry points (no GPU needed).
Mirrors ResourceManager::loadSplatsRaw + split_to_splat (reference
ResourceManager.cpp:36-65, Splat.h:60-91)."""
import ctypes as C

import numpy as np


def raw_to_splats(pkg, raw, center=False, lib=None):
    """(n,32) uint8 SplatRaw records -> (n,20) float32 Splat records."""
    lib = lib or pkg.load_library()
    raw = np.ascontiguousarray(raw, np.uint8).reshape(-1, 32)
    out = np.empty((raw.shape[0], 20), np.float32)
    rc = lib.gsx_host_convert_raw(raw.ctypes.data_as(C.c_void_p), C.c_uint64(raw.shape[0]), C.c_int(1 if center else 0),
                                  out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise pkg.GsxError(rc, "gsx_host_convert_raw")
    return out


def load_splat_file(pkg, path, center=False, lib=None):
    """`.splat` file -> (n,20) float32 Splat records; a missing file gives n = 0 like the reference."""
    lib = lib or pkg.load_library()
    ptr, n = C.c_void_p(), C.c_uint64(0)
    rc = lib.gsx_host_load_splat_file(str(path).encode(), C.c_int(1 if center else 0), C.byref(ptr), C.byref(n))
    if rc != 0:
        raise pkg.GsxError(rc, "gsx_host_load_splat_file")
    if n.value == 0:
        return np.zeros((0, 20), np.float32)
    buf = (C.c_float * (n.value * 20)).from_address(ptr.value)
    out = np.frombuffer(buf, np.float32).reshape(-1, 20).copy()
    lib.gsx_host_free.argtypes = [C.c_void_p]
    lib.gsx_host_free(ptr)
    return out


def save_splat_file(path, raw):
    np.ascontiguousarray(raw, np.uint8).reshape(-1, 32).tofile(path)
