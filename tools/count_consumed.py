"""Debug experiment (needs a -DGSX_DEBUG_COUNT build of libgsx, see GSX_LIB): how many of the emitted
(tile, splat) pairs the front-to-back rasterizer consumes before its tiles saturate, per tile."""
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
    g = pkg.Gsx(0); g.upload(sp); g.set_profiling(True)
    w = cam.orbit_pose(5.0, []); v, p, pos = cam.camera_matrices(w, 45.0, W / H)
    g.sort(pos, download=False)
    img = g.render(pkg.make_uniforms(p, v), W, H, download=True)
    t = g.timings()
    px = np.ascontiguousarray(img).view(np.uint32).reshape(H, W)
    cons = px[0::16, 0::16].astype(np.int64); tot = px[0::16, 1::16].astype(np.int64)
    print(name, "pairs", t["tile_pairs"], "listed", tot.sum(), "consumed", cons.sum(), "raster_ms", t["raster_ms"], flush=True)
    for q in (50, 90, 99, 99.9, 100):
        print("  pct", q, "consumed", np.percentile(cons, q), "listed", np.percentile(tot, q))
    idx = np.argsort(cons.ravel())[::-1][:12]
    print("  top tiles (ty,tx,consumed,listed):", [(int(i // cons.shape[1]), int(i % cons.shape[1]), int(cons.ravel()[i]), int(tot.ravel()[i])) for i in idx])
    print("  consumed per tile row (mean):", cons.mean(axis=1).astype(int)[::8])
    print("  consumed per tile col (mean):", cons.mean(axis=0).astype(int)[::8])
    g.close()
