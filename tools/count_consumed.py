import ctypes as C, sys, os, importlib
sys.path.insert(0,"tests"); sys.path.insert(0,"oracle")
from conftest import load_pkg
pkg=load_pkg(); synth=importlib.import_module("gaussian_splatting_amd.synth"); scene=importlib.import_module("gaussian_splatting_amd.scene"); cam=importlib.import_module("gaussian_splatting_amd.camera")
import numpy as np
for name in ("clustered","uniform","overlap"):
    raw=synth.SCENES[name](6_000_000); sp=scene.raw_to_splats(pkg,raw)
    g=pkg.Gsx(0); g.upload(sp)
    w=cam.orbit_pose(5.0,[]); v,p,pos=cam.camera_matrices(w,45.0,1920/1080)
    g.sort(pos,download=False); g.render(pkg.make_uniforms(p,v),1920,1080,download=False); g.sync()
    out=(C.c_uint32*8)(); g.lib.gsx_debug_flags(g.ctx,out)
    t=g.timings()
    print(name,"pairs",t["tile_pairs"],"consumed",out[5], "frac", out[5]/max(t["tile_pairs"],1), flush=True)
    g.close()
