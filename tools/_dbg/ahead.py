import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import numpy as np
import conftest, orc
pkg = conftest.load_pkg()
import importlib
synth = importlib.import_module("gaussian_splatting_amd.synth")
import test_gpu_render as T
oracle = orc.Oracle()
g = pkg.Gsx(0)
w, h = 240, 176
splats = T.dense_scene(oracle, synth, 20000, seed=7)
g.upload(splats)
cams = [T.camera(oracle, w, h, steps=s) for s in (0, 3, 6, 9, 12)]
unis = [pkg.make_uniforms(proj, view) for view, proj, _ in cams]
ref = []
for (view, proj, cam), u in zip(cams, unis):
    g.sort(cam, download=False)
    ref.append(g.render(u, w, h).copy())
g.set_option("sort_ahead", 1)
g.sort(cams[0][2], download=False)
for k in range(5):
    if k + 1 < 5:
        g.sort(cams[k + 1][2], download=False)
    g.render(unis[k], w, h, download=False)
    g.sync()
    ptr, fw, fh = g.framebuffer_dev()
    img = T._read_device(ptr, w * h * 4).reshape(h, w, 4)
    print(k, [bool(np.array_equal(img, r)) for r in ref], int((img != ref[k]).sum()))
