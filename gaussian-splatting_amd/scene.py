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


def load_ply_raw(pkg, path, lib=None):
    """3DGS `.ply` -> (n,32) uint8 SplatRaw records (include/gsx.h gsx_host_load_ply_raw)."""
    lib = lib or pkg.load_library()
    ptr, n = C.c_void_p(), C.c_uint64(0)
    rc = lib.gsx_host_load_ply_raw(str(path).encode(), C.byref(ptr), C.byref(n))
    if rc != 0:
        raise pkg.GsxError(rc, "gsx_host_load_ply_raw")
    if n.value == 0:
        return np.zeros((0, 32), np.uint8)
    out = np.frombuffer((C.c_uint8 * (n.value * 32)).from_address(ptr.value), np.uint8).reshape(-1, 32).copy()
    lib.gsx_host_free.argtypes = [C.c_void_p]
    lib.gsx_host_free(ptr)
    return out


def load_ply_file(pkg, path, center=False, lib=None):
    """3DGS `.ply` -> (n,20) float32 Splat records ready for Gsx.upload."""
    lib = lib or pkg.load_library()
    ptr, n = C.c_void_p(), C.c_uint64(0)
    rc = lib.gsx_host_load_ply_file(str(path).encode(), C.c_int(1 if center else 0), C.byref(ptr), C.byref(n))
    if rc != 0:
        raise pkg.GsxError(rc, "gsx_host_load_ply_file")
    if n.value == 0:
        return np.zeros((0, 20), np.float32)
    out = np.frombuffer((C.c_float * (n.value * 20)).from_address(ptr.value), np.float32).reshape(-1, 20).copy()
    lib.gsx_host_free.argtypes = [C.c_void_p]
    lib.gsx_host_free(ptr)
    return out


def save_ply_file(path, xyz, f_dc, opacity, log_scale, rot, extra_rest=0, order=None):
    """Write a 3DGS-style binary little-endian PLY (test/tool helper): float properties x y z nx ny nz f_dc_0..2
    f_rest_* opacity scale_0..2 rot_0..3, or in `order` if given."""
    n = len(xyz)
    cols = {"x": xyz[:, 0], "y": xyz[:, 1], "z": xyz[:, 2], "nx": np.zeros(n), "ny": np.zeros(n), "nz": np.zeros(n),
            "f_dc_0": f_dc[:, 0], "f_dc_1": f_dc[:, 1], "f_dc_2": f_dc[:, 2], "opacity": opacity,
            "scale_0": log_scale[:, 0], "scale_1": log_scale[:, 1], "scale_2": log_scale[:, 2],
            "rot_0": rot[:, 0], "rot_1": rot[:, 1], "rot_2": rot[:, 2], "rot_3": rot[:, 3]}
    for k in range(extra_rest):
        cols[f"f_rest_{k}"] = np.full(n, 0.01 * k)
    names = order or (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{k}" for k in range(extra_rest)] +
                      ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])
    head = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join(f"property float {m}\n" for m in names) + "end_header\n"
    data = np.stack([np.asarray(cols[m], np.float32) for m in names], axis=1)
    with open(path, "wb") as f:
        f.write(head.encode())
        f.write(np.ascontiguousarray(data, "<f4").tobytes())
