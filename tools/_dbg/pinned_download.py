"""How much of a synchronous gsx_render(..., out_host) is the pageable device-to-host copy?  Same frames into a numpy array and into
pinned host memory (a torch pinned tensor viewed as numpy)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import load_pkg, orbit_path
import torch
pkg = load_pkg()
synth = importlib.import_module("gaussian_splatting_amd.synth")
scene = importlib.import_module("gaussian_splatting_amd.scene")
W, H = 1920, 1080
splats = scene.raw_to_splats(pkg, synth.clustered_cloud(6_000_000, seed=6), center=False)
path = orbit_path(pkg, 120, W, H)
g = pkg.Gsx(0)
g.upload(splats)
pag = np.zeros((H, W, 4), np.uint8)
pin_t = torch.zeros((H, W, 4), dtype=torch.uint8).pin_memory()
pin = pin_t.numpy()
for name, out in (("pageable", pag), ("pinned", pin), ("pageable", pag), ("pinned", pin), ("none", None)):
    for rep in range(2):
        t0 = time.perf_counter()
        for k in range(40):
            p = path[k]
            g.sort(p[2], download=False)
            if out is None:
                g.render(pkg.make_uniforms(p[1], p[0]), W, H, download=False); g.sync()
            else:
                g.render(pkg.make_uniforms(p[1], p[0]), W, H, out=out)
        dt = (time.perf_counter() - t0) / 40 * 1e3
    print(f"{name:9s} {dt:.3f} ms per frame (sort + render + download, synchronous)")
assert np.array_equal(pag, pin)
g.close()
