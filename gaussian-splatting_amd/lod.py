"""Host-side LOD structures (Octree / HC / GridHC) through libgsx's host entry points.
Construction follows the reference's load-time code (include/gsx.h, host/gsx_lod_host.hpp)."""
import ctypes as C
import sys

import numpy as np


def _lib():
    pkg = sys.modules["gaussian_splatting_amd"]
    if not hasattr(_lib, "lib"):
        L = _lib.lib = pkg.load_library()
        for f in ("gsx_host_lod_size", "gsx_host_lod_roots", "gsx_host_lod_cells"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        L.gsx_host_lod_free.argtypes = [C.c_void_p]
    return _lib.lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HostLod:
    def __init__(self, handle, kind):
        self.h, self.kind = handle, kind
        L = _lib()
        self.n = L.gsx_host_lod_size(self.h)
        self.splats = np.empty((self.n, 20), np.float32)
        assert L.gsx_host_lod_splats(self.h, _p(self.splats)) == 0
        if kind == "octree":
            self.center3, self.half3 = np.empty((self.n, 3), np.float32), np.empty((self.n, 3), np.float32)
            self.parent, self.is_leaf = np.empty(self.n, np.uint32), np.empty(self.n, np.uint8)
            self.first_child, self.n_children = np.empty(self.n, np.uint32), np.empty(self.n, np.uint32)
            assert L.gsx_host_lod_flat_octree(self.h, _p(self.center3), _p(self.half3), _p(self.parent), _p(self.is_leaf),
                                              _p(self.first_child), _p(self.n_children)) == 0
        else:
            r, c = L.gsx_host_lod_roots(self.h), L.gsx_host_lod_cells(self.h)
            self.parent, self.is_leaf = np.empty(self.n, np.uint32), np.empty(self.n, np.uint8)
            self.error, self.weight = np.empty(self.n, np.float32), np.empty(self.n, np.float32)
            self.child2, self.roots = np.empty(2 * self.n, np.uint32), np.empty(r, np.uint32)
            self.cell_root_start = np.empty(c + 1, np.uint32) if kind == "gridhc" else None
            assert L.gsx_host_lod_flat_hc(self.h, _p(self.parent), _p(self.is_leaf), _p(self.error), _p(self.weight),
                                          _p(self.child2), _p(self.roots), _p(self.cell_root_start)) == 0

    def cut_depth(self, depth):
        out = np.empty(max(self.n, 1), np.uint32)
        m = C.c_uint64(0)
        assert _lib().gsx_host_lod_cut_depth(self.h, C.c_uint32(depth), _p(out), C.c_uint64(len(out)), C.byref(m)) == 0
        return out[:m.value].copy()

    def cut_camera(self, world, threshold, fov=45.0, aspect=1.0, near=0.1, far=100.0):
        out = np.empty(max(self.n, 1), np.uint32)
        m = C.c_uint64(0)
        w = np.ascontiguousarray(world, np.float32)
        assert _lib().gsx_host_lod_cut_camera(self.h, _p(w), C.c_float(fov), C.c_float(aspect), C.c_float(near), C.c_float(far),
                                              C.c_float(threshold), _p(out), C.c_uint64(len(out)), C.byref(m)) == 0
        return out[:m.value].copy()

    def upload(self, gsx):
        """Scene + flattened tree into a device context."""
        gsx.upload(self.splats)
        if self.kind == "octree":
            gsx.set_lod_octree(self.center3, self.half3, self.parent, self.is_leaf)
        else:
            gsx.set_lod_hc(self.parent, self.is_leaf, self.error, self.weight)

    def close(self):
        if self.h:
            _lib().gsx_host_lod_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_octree(raw, center=False):
    raw = np.ascontiguousarray(raw, np.uint8).reshape(-1, 32)
    h = C.c_void_p()
    rc = _lib().gsx_host_octree_build(_p(raw), C.c_uint64(raw.shape[0]), C.c_int(1 if center else 0), C.byref(h))
    assert rc == 0, rc
    return HostLod(h, "octree")


def build_hc(splats):
    splats = np.ascontiguousarray(splats, np.float32).reshape(-1, 20)
    h = C.c_void_p()
    rc = _lib().gsx_host_hc_build(_p(splats), C.c_uint64(splats.shape[0]), C.byref(h))
    assert rc == 0, rc
    return HostLod(h, "hc")


def build_gridhc(splats, sub=(20, 20, 20)):
    splats = np.ascontiguousarray(splats, np.float32).reshape(-1, 20)
    h = C.c_void_p()
    rc = _lib().gsx_host_gridhc_build(_p(splats), C.c_uint64(splats.shape[0]), *[C.c_uint32(s) for s in sub], C.byref(h))
    assert rc == 0, rc
    return HostLod(h, "gridhc")
