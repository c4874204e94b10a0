"""Debug experiment (needs a -DGSX_DEBUG_COUNT build of libgsx, see GSX_LIB): rasterizer work counters."""
import ctypes as C, sys, os, importlib
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from conftest import load_pkg
pkg = load_pkg()
synth = importlib.import_module("gaussian_splatting_amd.synth")
scene = importlib.import_module("gaussian_splatting_amd.scene")
cam = importlib.import_module("gaussian_splatting_amd.camera")
import numpy as np
W, H = 1920, 1080
for name in sys.argv[1:] or ("clustered",):
    raw = synth.SCENES[name](6_000_000); sp = scene.raw_to_splats(pkg, raw)
    g = pkg.Gsx(0); g.upload(sp)
    w = cam.orbit_pose(5.0, []); v, p, pos = cam.camera_matrices(w, 45.0, W / H)
    g.sort(pos, download=False)
    g.render(pkg.make_uniforms(p, v), W, H, download=False); g.sync()
    a = np.zeros(17 + 1 + 8, np.uint32)
    assert g.lib.gsx_debug_read(g.ctx, 8, a.ctypes.data_as(C.c_void_p), C.c_uint64(a.nbytes)) == 0
    d = a[18:].view(np.uint64)
    t = g.timings()
    print(name, "pairs", t["tile_pairs"], "listed(tile visits)", d[3] // 2, "wave-splat shades", d[0], "hit lanes", d[1], "useful lanes", d[2],
          "lane efficiency", d[1] / (64.0 * d[0]), "useful", d[2] / (64.0 * d[0]), "shades per listed pair", d[0] / (d[3] / 2))
    g.close()
