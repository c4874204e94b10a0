"""gaussian-splatting_amd — host-side Python plumbing over libgsx.so.

The product is the C-ABI library built from csrc/ (include/gsx.h); this module is
a thin ctypes mirror used by tests and bench.py.  It never falls back to a CPU
path: if libgsx.so is missing the import of `Gsx` raises.

The directory name contains a hyphen (it mirrors the reference repo's name), so
load it with `gsx_load()` from `__graft_entry__` / `tests/conftest.py` or via
importlib; `import gaussian-splatting_amd` is not valid Python.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSX_LIB") or os.path.join(_HERE, "libgsx.so")

GSX_OK, GSX_E_INVALID, GSX_E_HIP, GSX_E_NOMEM, GSX_E_STATE, GSX_E_UNSUPPORTED, GSX_E_INTERNAL = 0, -1, -2, -3, -4, -5, -6

# every symbol include/gsx.h declares
ABI_SYMBOLS = [
    "gsx_create", "gsx_destroy", "gsx_last_error", "gsx_sync", "gsx_set_profiling", "gsx_get_timings", "gsx_get_scatter_samples", "gsx_get_sort_samples", "gsx_set_option",
    "gsx_upload_splats", "gsx_upload_splats_dev", "gsx_select_all", "gsx_select_list", "gsx_set_lod_octree",
    "gsx_set_lod_hc", "gsx_select_octree", "gsx_select_hc", "gsx_select_band", "gsx_get_selection", "gsx_sort", "gsx_sort_noalloc", "gsx_get_sorted_dev",
    "gsx_get_keys", "gsx_render", "gsx_get_framebuffer_dev", "gsx_set_framebuffer_dev", "gsx_get_stream",
    "gsx_comm_unique_id", "gsx_comm_init", "gsx_comm_destroy", "gsx_gather",
    "gsx_host_convert_raw", "gsx_host_load_splat_file", "gsx_host_free", "gsx_host_alloc_pinned", "gsx_host_free_pinned", "gsx_host_load_ply_raw", "gsx_host_load_ply_file", "gsx_host_camera", "gsx_host_orbit_pose", "gsx_host_orbit_events",
    "gsx_host_octree_build", "gsx_host_hc_build", "gsx_host_gridhc_build", "gsx_host_lod_free", "gsx_host_lod_size",
    "gsx_host_lod_roots", "gsx_host_lod_cells", "gsx_host_lod_splats", "gsx_host_lod_flat_octree", "gsx_host_lod_flat_hc",
    "gsx_host_lod_cut_depth", "gsx_host_lod_cut_camera",
]


class Uniforms(C.Structure):
    """gsx_uniforms (reference main.cpp:48-58), 224 bytes."""
    _fields_ = [("projectionMatrix", C.c_float * 16), ("viewMatrix", C.c_float * 16), ("modelMatrix", C.c_float * 16),
                ("splatSize", C.c_float), ("cutOff", C.c_float), ("fov", C.c_float), ("bayerSize", C.c_uint32),
                ("bayerScale", C.c_float), ("padding", C.c_float * 3)]


class OctreeFlat(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("center3", C.c_void_p), ("half3", C.c_void_p), ("parent", C.c_void_p),
                ("is_leaf", C.c_void_p)]


class HcFlat(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("parent", C.c_void_p), ("is_leaf", C.c_void_p), ("error", C.c_void_p),
                ("weight", C.c_void_p)]


class Timings(C.Structure):
    _fields_ = [("select_ms", C.c_float), ("keygen_ms", C.c_float), ("sort_ms", C.c_float), ("scatter_ms", C.c_float), ("project_ms", C.c_float),
                ("bin_ms", C.c_float), ("raster_ms", C.c_float), ("total_ms", C.c_float), ("sort_passes", C.c_uint32),
                ("sort_launches", C.c_uint32), ("tile_pairs", C.c_uint64), ("render_iters", C.c_uint32), ("scatter_sample", C.c_uint32),
                ("raster_shades", C.c_uint64), ("raster_blends", C.c_uint64), ("raster_staged", C.c_uint64), ("render_capped", C.c_uint32), ("render_redos", C.c_uint32),
                ("sort_form", C.c_uint32), ("reserved0", C.c_uint32)]


assert C.sizeof(Uniforms) == 224


def make_uniforms(proj, view, model=None, splatSize=1.0, cutOff=3.5, fov=45.0, bayerSize=1, bayerScale=1.0):
    u = Uniforms()
    model = np.eye(4, dtype=np.float32).reshape(-1) if model is None else model
    for name, m in (("projectionMatrix", proj), ("viewMatrix", view), ("modelMatrix", model)):
        arr = np.ascontiguousarray(m, dtype=np.float32).reshape(-1)
        C.memmove(getattr(u, name), arr.ctypes.data, 64)
    u.splatSize, u.cutOff, u.fov, u.bayerSize, u.bayerScale = splatSize, cutOff, fov, bayerSize, bayerScale
    return u


class GsxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gsx error {code}: {msg}")
        self.code = code


def load_library(path=None):
    """dlopen libgsx.so.  Raises (never falls back) when the HIP library is absent."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(path)
    lib.gsx_last_error.restype = C.c_char_p
    lib.gsx_last_error.argtypes = [C.c_void_p]
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pinned_frame(height, width, lib=None):
    """An (height, width, 4) uint8 array in page-locked host memory (gsx_host_alloc_pinned) to pass as Gsx.render(..., out=):
    the read-back is then one DMA into the array (6 M splats, 1080p: 1.23 -> 0.91 ms per synchronous frame).  The memory is
    released when the array and every view of it are gone."""
    import weakref
    lib = lib or load_library()
    lib.gsx_host_alloc_pinned.restype = C.c_void_p
    lib.gsx_host_alloc_pinned.argtypes = [C.c_uint64]
    lib.gsx_host_free_pinned.argtypes = [C.c_void_p]
    nbytes = int(height) * int(width) * 4
    p = lib.gsx_host_alloc_pinned(max(nbytes, 1))
    if not p:
        raise MemoryError("gsx_host_alloc_pinned failed")
    buf = (C.c_uint8 * max(nbytes, 1)).from_address(p)
    weakref.finalize(buf, lib.gsx_host_free_pinned, C.c_void_p(p))     # numpy keeps `buf` alive as the array's base
    a = np.frombuffer(buf, np.uint8, count=nbytes).reshape(int(height), int(width), 4)
    a[...] = 0
    return a


class Gsx:
    """One context per GPU (include/gsx.h)."""

    def __init__(self, device=0, lib=None):
        self.lib = lib or load_library()
        self.ctx = C.c_void_p()
        rc = self.lib.gsx_create(C.c_int(device), C.byref(self.ctx))
        if rc != GSX_OK:
            raise GsxError(rc, "gsx_create failed (no usable HIP device?)")
        self.n = 0
        self._keep = []

    def _chk(self, rc):
        if rc != GSX_OK:
            raise GsxError(rc, self.lib.gsx_last_error(self.ctx).decode())

    def close(self):
        if self.ctx:
            self.comm_destroy()
            self.lib.gsx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- scene --
    def upload(self, splats):
        splats = np.ascontiguousarray(splats, np.float32).reshape(-1, 20)
        self._chk(self.lib.gsx_upload_splats(self.ctx, _ptr(splats), C.c_uint64(splats.shape[0])))
        self.n = splats.shape[0]

    def upload_dev(self, dev_ptr, n):
        self._chk(self.lib.gsx_upload_splats_dev(self.ctx, C.c_void_p(dev_ptr), C.c_uint64(n)))
        self.n = n

    # -- selection --
    def select_all(self):
        self._chk(self.lib.gsx_select_all(self.ctx))

    def select_list(self, idx):
        idx = np.ascontiguousarray(idx, np.uint32)
        self._chk(self.lib.gsx_select_list(self.ctx, _ptr(idx), C.c_uint32(len(idx))))

    def set_lod_octree(self, center3, half3, parent, is_leaf):
        arrs = [np.ascontiguousarray(center3, np.float32), np.ascontiguousarray(half3, np.float32),
                np.ascontiguousarray(parent, np.uint32), np.ascontiguousarray(is_leaf, np.uint8)]
        t = OctreeFlat(len(arrs[2]), *[a.ctypes.data for a in arrs])
        self._chk(self.lib.gsx_set_lod_octree(self.ctx, C.byref(t)))

    def set_lod_hc(self, parent, is_leaf, error, weight):
        arrs = [np.ascontiguousarray(parent, np.uint32), np.ascontiguousarray(is_leaf, np.uint8),
                np.ascontiguousarray(error, np.float32), np.ascontiguousarray(weight, np.float32)]
        t = HcFlat(len(arrs[0]), *[a.ctypes.data for a in arrs])
        self._chk(self.lib.gsx_set_lod_hc(self.ctx, C.byref(t)))

    def select_octree(self, viewproj, min_screen_area, want_count=True):
        vp = np.ascontiguousarray(viewproj, np.float32)
        m = C.c_uint32(0)
        self._chk(self.lib.gsx_select_octree(self.ctx, _ptr(vp), C.c_float(min_screen_area), C.byref(m) if want_count else None))
        return m.value

    def select_hc(self, cam_pos, threshold, want_count=True):
        cam = np.ascontiguousarray(cam_pos, np.float32)
        m = C.c_uint32(0)
        self._chk(self.lib.gsx_select_hc(self.ctx, _ptr(cam), C.c_float(threshold), C.byref(m) if want_count else None))
        return m.value

    def select_band(self, uniforms, width, height, y0, y1, want_count=True):
        """gsx_select_band: the splats whose quad can reach rows [y0, y1) (tile-band sharding)."""
        m = C.c_uint32(0)
        self._chk(self.lib.gsx_select_band(self.ctx, C.byref(uniforms), C.c_uint32(width), C.c_uint32(height), C.c_uint32(y0), C.c_uint32(y1),
                                           C.byref(m) if want_count else None))
        return m.value

    def selection(self):
        m = C.c_uint32(0)
        self._chk(self.lib.gsx_get_selection(self.ctx, None, C.c_uint32(0), C.byref(m)))
        out = np.empty(m.value, np.uint32)
        self._chk(self.lib.gsx_get_selection(self.ctx, _ptr(out), C.c_uint32(len(out)), C.byref(m)))
        return out

    # -- sort --
    def sort(self, cam_pos, download=True):
        cam = np.ascontiguousarray(cam_pos, np.float32)
        if not download:
            self._chk(self.lib.gsx_sort(self.ctx, _ptr(cam), None))
            return None
        out = np.empty(max(self.n, 1), np.uint32)
        self._chk(self.lib.gsx_sort(self.ctx, _ptr(cam), _ptr(out)))
        dev = C.c_void_p()
        m = C.c_uint32(0)
        self._chk(self.lib.gsx_get_sorted_dev(self.ctx, C.byref(dev), C.byref(m)))
        return out[:m.value].copy()

    def sort_noalloc(self, cam_pos, out_dev_ptr, capacity):
        """gsx_sort_noalloc: queue the sort of the current selection into caller-owned device memory (no allocation, no wait)."""
        cam = np.ascontiguousarray(cam_pos, np.float32)
        self._chk(self.lib.gsx_sort_noalloc(self.ctx, _ptr(cam), C.c_void_p(out_dev_ptr), C.c_uint32(int(capacity))))

    def sorted_dev(self):
        dev = C.c_void_p()
        m = C.c_uint32(0)
        self._chk(self.lib.gsx_get_sorted_dev(self.ctx, C.byref(dev), C.byref(m)))
        return dev.value, m.value

    def keys(self, cam_pos):
        cam = np.ascontiguousarray(cam_pos, np.float32)
        out = np.empty(max(self.n, 1), np.float32)
        self._chk(self.lib.gsx_get_keys(self.ctx, _ptr(cam), _ptr(out), C.c_uint32(len(out))))
        m = C.c_uint32(0)
        self._chk(self.lib.gsx_get_selection(self.ctx, None, C.c_uint32(0), C.byref(m)))
        return out[:m.value].copy()

    # -- render --
    def render(self, uniforms, width, height, y0=0, y1=None, download=True, out=None):
        y1 = height if y1 is None else y1
        if download and out is None:
            out = np.zeros((height, width, 4), np.uint8)
        self._chk(self.lib.gsx_render(self.ctx, C.byref(uniforms), C.c_uint32(width), C.c_uint32(height), C.c_uint32(y0),
                                      C.c_uint32(y1), _ptr(out) if download else None))
        return out

    def framebuffer_dev(self):
        dev = C.c_void_p()
        w, h = C.c_uint32(0), C.c_uint32(0)
        self._chk(self.lib.gsx_get_framebuffer_dev(self.ctx, C.byref(dev), C.byref(w), C.byref(h)))
        return dev.value, w.value, h.value

    def set_framebuffer_dev(self, dev_ptr, capacity_bytes=0):
        """Render the following frames into caller-owned device memory (None = the library's own framebuffer);
        completes the queued frame into the previous target first."""
        self._chk(self.lib.gsx_set_framebuffer_dev(self.ctx, C.c_void_p(dev_ptr), C.c_uint64(int(capacity_bytes))))

    def stream(self):
        """hipStream_t (as an integer) every call of this context is queued on, e.g. for torch.cuda.ExternalStream."""
        st = C.c_void_p()
        self._chk(self.lib.gsx_get_stream(self.ctx, C.byref(st)))
        return st.value or 0

    # -- multi-GPU gather (include/gsx.h: gsx_comm_*, gsx_gather) --
    def comm_unique_id(self):
        """128-byte RCCL id (rank 0 creates it and hands it to the other ranks)."""
        buf = np.zeros(128, np.uint8)
        rc = self.lib.gsx_comm_unique_id(_ptr(buf))
        if rc != GSX_OK:
            raise GsxError(rc, "gsx_comm_unique_id failed (librccl missing?)")
        return buf

    def comm_init(self, id128, rank, nranks):
        """Collective: joins the communicator of `nranks` processes.  Kept in self.comm until close()."""
        comm = C.c_void_p()
        idp = None if id128 is None else _ptr(np.ascontiguousarray(id128, np.uint8))
        self._chk(self.lib.gsx_comm_init(self.ctx, idp, C.c_int(rank), C.c_int(nranks), C.byref(comm)))
        self.comm, self.comm_rank, self.comm_size = comm, rank, nranks
        return comm

    def comm_destroy(self):
        if getattr(self, "comm", None):
            self.lib.gsx_comm_destroy(self.comm)
            self.comm = None

    def gather(self, send_dev_ptr, send_bytes, recv_dev_ptr=None, recv_offsets=None, recv_bytes=None, root=0):
        """gsx_gather on this context's stream (device pointers as integers)."""
        off = None if recv_offsets is None else np.ascontiguousarray(recv_offsets, np.uint64)
        cnt = None if recv_bytes is None else np.ascontiguousarray(recv_bytes, np.uint64)
        self._chk(self.lib.gsx_gather(self.ctx, self.comm, C.c_void_p(send_dev_ptr), C.c_uint64(int(send_bytes)),
                                      C.c_void_p(recv_dev_ptr), _ptr(off), _ptr(cnt), C.c_int(root)))

    # -- misc --
    def sync(self):
        self._chk(self.lib.gsx_sync(self.ctx))

    def set_profiling(self, on=True):
        self._chk(self.lib.gsx_set_profiling(self.ctx, C.c_int(int(on))))   # 0 off, 1/True all stages, 2 sampled

    def set_option(self, name, value):
        """Tuning knob of the frame loop (include/gsx.h gsx_set_option): pair_budget, chunk_first, chunk_growth."""
        self._chk(self.lib.gsx_set_option(self.ctx, name.encode(), C.c_uint64(int(value))))

    def scatter_samples(self, reset=True):
        """Profiling level 2: ({pass: mean launch ms}, {pass: launches}) of the sampled depth-sort scatter launches."""
        sums = (C.c_double * 4)()
        cnt = (C.c_uint32 * 4)()
        self._chk(self.lib.gsx_get_scatter_samples(self.ctx, sums, cnt, C.c_int(1 if reset else 0)))
        return ({p: sums[p] / cnt[p] for p in range(4) if cnt[p]}, {p: int(cnt[p]) for p in range(4) if cnt[p]})

    def sort_samples(self, reset=True):
        """Profiling level 2: (mean ms of keys + sort as a whole, number of sampled sorts)."""
        s, n = C.c_double(0.0), C.c_uint32(0)
        self._chk(self.lib.gsx_get_sort_samples(self.ctx, C.byref(s), C.byref(n), C.c_int(1 if reset else 0)))
        return (s.value / n.value if n.value else 0.0), int(n.value)

    def timings(self):
        t = Timings()
        self._chk(self.lib.gsx_get_timings(self.ctx, C.byref(t)))
        return {f[0]: getattr(t, f[0]) for f in Timings._fields_}
