"""ctypes bindings for the CPU oracle (oracle/liborc.so) and, when present, the
compiled reference (oracle/_ref/libgsxref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


class Uniforms(C.Structure):
    """main.cpp:48-58 (224 bytes)."""
    _fields_ = [("proj", C.c_float * 16), ("view", C.c_float * 16), ("model", C.c_float * 16),
                ("splatSize", C.c_float), ("cutOff", C.c_float), ("fov", C.c_float),
                ("bayerSize", C.c_uint32), ("bayerScale", C.c_float), ("pad", C.c_float * 3)]


assert C.sizeof(Uniforms) == 224


def make_uniforms(proj, view, model=None, splatSize=1.0, cutOff=3.5, fov=45.0, bayerSize=1, bayerScale=1.0):
    u = Uniforms()
    model = np.eye(4, dtype=np.float32).reshape(-1) if model is None else model
    for name, m in (("proj", proj), ("view", view), ("model", model)):
        arr = np.ascontiguousarray(m, dtype=np.float32).reshape(-1)
        C.memmove(getattr(u, name), arr.ctypes.data, 64)
    u.splatSize, u.cutOff, u.fov, u.bayerSize, u.bayerScale = splatSize, cutOff, fov, bayerSize, bayerScale
    return u


def _opt(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, path=None):
        self.lib = L = C.CDLL(path or os.path.join(_HERE, "liborc.so"))
        L.orc_sort_reference_port.restype = C.c_double
        L.orc_bb_screen_area.restype = C.c_float
        L.orc_splat_weight.restype = C.c_float
        for f in ("orc_octree_cut", "orc_hc_cut_depth", "orc_hc_cut_metric"):
            getattr(L, f).restype = C.c_uint64

    # -- data model --
    def raw_to_split(self, raw):
        raw = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1, 32)
        out = np.empty((raw.shape[0], 14), np.float32)
        self.lib.orc_raw_to_split(_opt(raw), C.c_uint64(raw.shape[0]), _opt(out))
        return out

    def center_split(self, split):
        split = np.ascontiguousarray(split, np.float32).copy()
        self.lib.orc_center_split(_opt(split), C.c_uint64(split.shape[0]))
        return split

    def split_to_splat(self, split):
        split = np.ascontiguousarray(split, np.float32)
        out = np.empty((split.shape[0], 20), np.float32)
        self.lib.orc_split_to_splat(_opt(split), C.c_uint64(split.shape[0]), _opt(out))
        return out

    def raw_to_splat(self, raw, center=False):
        s = self.raw_to_split(raw)
        if center:
            s = self.center_split(s)
        return self.split_to_splat(s)

    # -- sort --
    def distance_keys(self, splats, cam, idx=None):
        n = splats.shape[0]
        m = n if idx is None else len(idx)
        out = np.zeros(n, np.float32)
        cam = np.ascontiguousarray(cam, np.float32)
        self.lib.orc_distance_keys(_opt(splats), C.c_uint64(n), _opt(idx), C.c_uint64(m), _opt(cam), _opt(out))
        return out

    def sort_contract(self, splats, cam, idx=None):
        n = splats.shape[0]
        m = n if idx is None else len(idx)
        out = np.empty(m, np.uint32)
        cam = np.ascontiguousarray(cam, np.float32)
        self.lib.orc_sort_contract(_opt(splats), C.c_uint64(n), _opt(idx), C.c_uint64(m), _opt(cam), _opt(out))
        return out

    def sort_reference_port(self, splats, cam, idx):
        """In-place on idx; returns seconds."""
        cam = np.ascontiguousarray(cam, np.float32)
        return self.lib.orc_sort_reference_port(_opt(splats), C.c_uint64(splats.shape[0]), _opt(idx),
                                                C.c_uint64(len(idx)), _opt(cam))

    def sort_reference_port_parallel(self, splats, cam, idx, threads=8):
        """Multi-threaded port (liborc_par.so: libstdc++ parallel mode, the image has no TBB headers) of the reference's
        PARALLEL build, SplatMesh.h:97-103.  In-place on idx; returns seconds."""
        if not hasattr(self, "par"):
            self.par = C.CDLL(os.path.join(_HERE, "liborc_par.so"))
            self.par.orc_sort_reference_port_parallel.restype = C.c_double
        cam = np.ascontiguousarray(cam, np.float32)
        return self.par.orc_sort_reference_port_parallel(_opt(splats), C.c_uint64(splats.shape[0]), _opt(idx),
                                                         C.c_uint64(len(idx)), _opt(cam), C.c_int(threads))

    # -- camera --
    def perspective(self, fov_deg=45.0, aspect=1.0, near=0.1, far=100.0):
        out = np.empty(16, np.float32)
        self.lib.orc_perspective(C.c_float(fov_deg), C.c_float(aspect), C.c_float(near), C.c_float(far), _opt(out))
        return out

    def inverse(self, m):
        out = np.empty(16, np.float32)
        self.lib.orc_mat4_inverse(_opt(np.ascontiguousarray(m, np.float32)), _opt(out))
        return out

    def mul(self, a, b):
        out = np.empty(16, np.float32)
        self.lib.orc_mat4_mul(_opt(np.ascontiguousarray(a, np.float32)), _opt(np.ascontiguousarray(b, np.float32)), _opt(out))
        return out

    def orbit_world(self, distance=5.0, axis=(0.0, 1.0, 0.0), angle=0.0, steps=0, rot3=None):
        rot = np.eye(3, dtype=np.float32).reshape(-1).copy() if rot3 is None else np.ascontiguousarray(rot3, np.float32).copy()
        ax = np.ascontiguousarray(axis, np.float32)
        for _ in range(steps):
            self.lib.orc_node_rotate(_opt(rot), _opt(ax), C.c_float(angle))
        out = np.empty(16, np.float32)
        self.lib.orc_orbit_world(_opt(rot), C.c_float(distance), _opt(out))
        return out, rot

    def camera(self, world, fov=45.0, aspect=1.0, near=0.1, far=100.0):
        """Returns (view, proj, viewproj, cam_pos) like Camera.h + BB.hpp:60-61."""
        view = self.inverse(world)
        proj = self.perspective(fov, aspect, near, far)
        return view, proj, self.mul(proj, view), np.array(world[12:15], np.float32)

    # -- LOD --
    def bb_screen_area(self, center, half, vp):
        return self.lib.orc_bb_screen_area(_opt(np.ascontiguousarray(center, np.float32)),
                                           _opt(np.ascontiguousarray(half, np.float32)),
                                           _opt(np.ascontiguousarray(vp, np.float32)))

    def octree_cut(self, center3, half3, first_child, n_children, vp, thr):
        n = len(n_children)
        out = np.empty(n, np.uint32)
        m = self.lib.orc_octree_cut(C.c_uint32(n), _opt(center3), _opt(half3), _opt(first_child), _opt(n_children),
                                    _opt(np.ascontiguousarray(vp, np.float32)), C.c_float(thr), _opt(out))
        return out[:m].copy()

    def hc_cut_depth(self, roots, child2, depth):
        out = np.empty(len(child2) // 2 + 1, np.uint32)
        m = self.lib.orc_hc_cut_depth(C.c_uint32(len(roots)), _opt(roots), _opt(child2), C.c_uint32(depth), _opt(out))
        return out[:m].copy()

    def hc_cut_metric(self, roots, child2, error, weight, pos3, cam, thr):
        out = np.empty(len(child2) // 2 + 1, np.uint32)
        m = self.lib.orc_hc_cut_metric(C.c_uint32(len(roots)), _opt(roots), _opt(child2), _opt(error), _opt(weight),
                                       _opt(np.ascontiguousarray(pos3, np.float32)),
                                       _opt(np.ascontiguousarray(cam, np.float32)), C.c_float(thr), _opt(out))
        return out[:m].copy()

    def splat_weight(self, splats):
        return np.array([self.lib.orc_splat_weight(_opt(s)) for s in np.ascontiguousarray(splats, np.float32)], np.float32)

    # -- raster --
    def vs_main(self, splat, u):
        out = np.empty(16, np.float32)
        self.lib.orc_vs_main(_opt(np.ascontiguousarray(splat, np.float32)), C.byref(u), _opt(out))
        return out.reshape(4, 4)

    def render(self, splats, order, u, w, h, y0=0, y1=None, want_f32=False):
        y1 = h if y1 is None else y1
        img = np.zeros((h, w, 4), np.uint8)
        fb = np.zeros((h, w, 4), np.float32) if want_f32 else None
        order = np.ascontiguousarray(order, np.uint32)
        self.lib.orc_render(_opt(splats), C.c_uint64(splats.shape[0]), _opt(order), C.c_uint64(len(order)),
                            C.byref(u), C.c_uint32(w), C.c_uint32(h), C.c_uint32(y0), C.c_uint32(y1),
                            _opt(img), _opt(fb))
        return (img, fb) if want_f32 else img


class Reference:
    """oracle/_ref/libgsxref.so — the reference's own sources, compiled."""

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libgsxref.so"))

    def __init__(self):
        self.lib = L = C.CDLL(os.path.join(_HERE, "_ref", "libgsxref.so"))
        L.ref_splat_weight.restype = C.c_float
        L.ref_bb_screen_area.restype = C.c_float
        for f in ("ref_octree_size", "ref_octree_get_indices", "ref_hc_size", "ref_hc_roots",
                  "ref_hc_get_indices_depth", "ref_hc_get_indices", "ref_gridhc_size", "ref_gridhc_roots",
                  "ref_gridhc_cells", "ref_gridhc_get_indices_error", "ref_gridhc_get_indices"):
            getattr(L, f).restype = C.c_uint64

    def raw_to_split(self, raw):
        raw = np.ascontiguousarray(raw, np.uint8).reshape(-1, 32)
        out = np.empty((raw.shape[0], 14), np.float32)
        self.lib.ref_raw_to_split(_opt(raw), C.c_uint64(raw.shape[0]), _opt(out))
        return out

    def split_to_splat(self, split):
        split = np.ascontiguousarray(split, np.float32)
        out = np.empty((split.shape[0], 20), np.float32)
        self.lib.ref_split_to_splat(_opt(split), C.c_uint64(split.shape[0]), _opt(out))
        return out

    def splat_weight(self, splats):
        return np.array([self.lib.ref_splat_weight(_opt(s)) for s in np.ascontiguousarray(splats, np.float32)], np.float32)

    def distance_keys(self, splats, cam, idx=None):
        n = splats.shape[0]
        m = n if idx is None else len(idx)
        out = np.zeros(n, np.float32)
        self.lib.ref_distance_keys(_opt(splats), C.c_uint64(n), _opt(idx), C.c_uint64(m),
                                   _opt(np.ascontiguousarray(cam, np.float32)), _opt(out))
        return out

    def sort(self, splats, cam, idx):
        idx = np.ascontiguousarray(idx, np.uint32).copy()
        self.lib.ref_sort(_opt(splats), C.c_uint64(splats.shape[0]), _opt(idx), C.c_uint64(len(idx)),
                          _opt(np.ascontiguousarray(cam, np.float32)))
        return idx

    def camera_matrices(self, world, fov=45.0, aspect=1.0, near=0.1, far=100.0):
        v, p, vp = (np.empty(16, np.float32) for _ in range(3))
        w = np.ascontiguousarray(world, np.float32)
        self.lib.ref_camera_matrices(_opt(w), C.c_float(fov), C.c_float(aspect), C.c_float(near), C.c_float(far), _opt(v), _opt(p))
        self.lib.ref_viewproj(_opt(w), C.c_float(fov), C.c_float(aspect), C.c_float(near), C.c_float(far), _opt(vp))
        return v, p, vp

    def orbit_world(self, distance=5.0, axis=(0.0, 1.0, 0.0), angle=0.0, steps=0):
        out = np.empty(16, np.float32)
        rot = np.empty(9, np.float32)
        self.lib.ref_orbit_world(C.c_float(distance), _opt(np.ascontiguousarray(axis, np.float32)), C.c_float(angle),
                                 C.c_uint32(steps), _opt(out), _opt(rot))
        return out, rot

    def orbit_events(self, events4, distance=5.0, orbit_rate=1.0, scroll_rate=1.0):
        """OrbitCamera.h:32-74 driven like main.cpp:139-155; rows {0,x,y,dt} cursor, {1,button,action,-}, {2,yoffset,-,dt} scroll."""
        ev = np.ascontiguousarray(events4, np.float64).reshape(-1, 4)
        out = np.empty(16, np.float32)
        self.lib.ref_orbit_events(C.c_float(distance), C.c_float(orbit_rate), C.c_float(scroll_rate), _opt(ev), C.c_uint32(len(ev)), _opt(out))
        return out

    def bb_screen_area(self, bb16, world, fov=45.0, aspect=1.0, near=0.1, far=100.0):
        return self.lib.ref_bb_screen_area(_opt(np.ascontiguousarray(bb16, np.float32)),
                                           _opt(np.ascontiguousarray(world, np.float32)), C.c_float(fov),
                                           C.c_float(aspect), C.c_float(near), C.c_float(far))

    # Octree
    def octree_build(self, split):
        split = np.ascontiguousarray(split, np.float32)
        h = self.lib.ref_octree_build(_opt(split), C.c_uint64(split.shape[0]))
        n = self.lib.ref_octree_size(h)
        d = dict(handle=h, splats=np.empty((n, 20), np.float32), bb16=np.empty((n, 16), np.float32),
                 center3=np.empty((n, 3), np.float32), half3=np.empty((n, 3), np.float32),
                 first_child=np.empty(n, np.uint32), n_children=np.empty(n, np.uint32),
                 node_splat=np.empty(n, np.uint32))
        self.lib.ref_octree_export(h, _opt(d["splats"]), _opt(d["bb16"]), _opt(d["center3"]), _opt(d["half3"]),
                                   _opt(d["first_child"]), _opt(d["n_children"]), _opt(d["node_splat"]))
        return d

    def octree_get_indices(self, h, world, thr, fov=45.0, aspect=1.0, near=0.1, far=100.0):
        out = np.empty(self.lib.ref_octree_size(h), np.uint32)
        m = self.lib.ref_octree_get_indices(h, _opt(np.ascontiguousarray(world, np.float32)), C.c_float(fov),
                                            C.c_float(aspect), C.c_float(near), C.c_float(far), C.c_float(thr), _opt(out))
        return out[:m].copy()

    # HC
    def hc_build(self, splats):
        splats = np.ascontiguousarray(splats, np.float32)
        h = self.lib.ref_hc_build(_opt(splats), C.c_uint64(splats.shape[0]))
        n, r = self.lib.ref_hc_size(h), self.lib.ref_hc_roots(h)
        d = dict(handle=h, splats=np.empty((n, 20), np.float32), child2=np.empty(2 * n, np.uint32),
                 error=np.empty(n, np.float32), roots=np.empty(r, np.uint32))
        self.lib.ref_hc_export(h, _opt(d["splats"]), _opt(d["child2"]), _opt(d["error"]), _opt(d["roots"]))
        return d

    def hc_get_indices_depth(self, h, depth):
        out = np.empty(self.lib.ref_hc_size(h), np.uint32)
        m = self.lib.ref_hc_get_indices_depth(h, C.c_uint32(depth), _opt(out))
        return out[:m].copy()

    def hc_get_indices(self, h, world, thr):
        out = np.empty(self.lib.ref_hc_size(h), np.uint32)
        m = self.lib.ref_hc_get_indices(h, _opt(np.ascontiguousarray(world, np.float32)), C.c_float(thr), _opt(out))
        return out[:m].copy()

    # GridHC
    def gridhc_build(self, splats, sub=(4, 4, 4)):
        splats = np.ascontiguousarray(splats, np.float32)
        h = self.lib.ref_gridhc_build(_opt(splats), C.c_uint64(splats.shape[0]), *[C.c_uint32(s) for s in sub])
        n, r, c = self.lib.ref_gridhc_size(h), self.lib.ref_gridhc_roots(h), self.lib.ref_gridhc_cells(h)
        d = dict(handle=h, splats=np.empty((n, 20), np.float32), child2=np.empty(2 * n, np.uint32),
                 error=np.empty(n, np.float32), roots=np.empty(r, np.uint32),
                 cell_root_start=np.empty(c + 1, np.uint32))
        self.lib.ref_gridhc_export(h, _opt(d["splats"]), _opt(d["child2"]), _opt(d["error"]), _opt(d["roots"]),
                                   _opt(d["cell_root_start"]))
        return d

    def gridhc_get_indices_error(self, h, depth, error=0.2):
        out = np.empty(self.lib.ref_gridhc_size(h), np.uint32)
        m = self.lib.ref_gridhc_get_indices_error(h, C.c_uint32(depth), C.c_float(error), _opt(out))
        return out[:m].copy()

    def gridhc_get_indices(self, h, world, thr):
        out = np.empty(self.lib.ref_gridhc_size(h), np.uint32)
        m = self.lib.ref_gridhc_get_indices(h, _opt(np.ascontiguousarray(world, np.float32)), C.c_float(thr), _opt(out))
        return out[:m].copy()

    def release(self, h):
        self.lib.ref_release(h)
