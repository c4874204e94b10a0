#!/usr/bin/env python3
"""Generates the fixtures in tests/golden/ by running the REFERENCE's own code.

Run in the build container only (needs /root/reference):
    make -C oracle            # builds oracle/_ref/libgsxref.so from the reference sources
    python tests/golden/make_golden.py

Every expected value below comes out of oracle/_ref/libgsxref.so, i.e. the
reference's Splat.h / BB.hpp / Camera.h / Node.cpp / Octree.cpp / HC.cpp /
GridHC.cpp and its vendored glm, compiled unmodified.  Inputs are synthetic
(gaussian-splatting_amd/synth.py) and stored next to the outputs, so the
fixtures are self-contained data (no reference source text).
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402

spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "gaussian-splatting_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synth)

OUT = os.path.dirname(os.path.abspath(__file__))
R = orc.Reference()
ASPECT = 16.0 / 9.0


def poses(n_steps_list, axis=(0.0, 1.0, 0.0), angle=np.float32(2 * np.pi / 120)):
    out = []
    for s in n_steps_list:
        w, rot = R.orbit_world(5.0, axis, angle, s)
        v, p, vp = R.camera_matrices(w, 45.0, ASPECT)
        out.append(dict(steps=s, world=w, rot=rot, view=v, proj=p, viewproj=vp))
    return out


def main():
    # ---- (1) data model: SplatRaw -> SplatSplit -> Splat ----
    raw = synth.uniform_cloud(240, seed=11)
    edge = np.zeros((16, 32), np.uint8)
    edge[:, 0:24] = np.tile(np.array([0.5, -1.25, 2.0, 0.01, 0.02, 0.03], np.float32).view(np.uint8), (16, 1))
    for i in range(16):  # extreme colour / quaternion bytes
        edge[i, 24:28] = [0, 255, 128, (17 * i) % 256]
        edge[i, 28:32] = [(255 if i & 1 else 0), (255 if i & 2 else 1), (255 if i & 4 else 127), (255 if i & 8 else 129)]
    raw = np.concatenate([raw, edge])
    split = R.raw_to_split(raw)
    splat = R.split_to_splat(split)
    np.savez_compressed(os.path.join(OUT, "datamodel.npz"), raw=raw, split=split, splat=splat,
                        weight=R.splat_weight(splat))

    # ---- (4) camera: 120-frame orbit about +Y (Node::rotate) ----
    ps = poses(range(0, 120))
    np.savez_compressed(os.path.join(OUT, "camera_orbit.npz"), world=np.stack([p["world"] for p in ps]),
                        rot=np.stack([p["rot"] for p in ps]), view=np.stack([p["view"] for p in ps]),
                        proj=np.stack([p["proj"] for p in ps]), viewproj=np.stack([p["viewproj"] for p in ps]),
                        angle=np.float32(2 * np.pi / 120), aspect=np.float32(ASPECT))

    # ---- (2) keys + permutations ----
    sort_cases = {}
    for name, raw_s in (("uniform1k", synth.uniform_cloud(1000, seed=1)), ("lattice12", synth.lattice_cloud(12, seed=3)),
                        ("clustered20k", synth.clustered_cloud(20000, seed=6))):
        sp = R.split_to_splat(R.raw_to_split(raw_s))
        for pi, p in enumerate(poses([0, 30, 77])):
            cam = p["world"][12:15].copy()
            if name == "lattice12":
                cam = np.array([0.0, 0.0, 5.0], np.float32) if pi == 0 else cam
            keys = R.distance_keys(sp, cam)
            perm = R.sort(sp, cam, np.arange(len(sp), dtype=np.uint32))
            sort_cases[f"{name}_p{pi}_cam"] = cam
            sort_cases[f"{name}_p{pi}_keys"] = keys
            sort_cases[f"{name}_p{pi}_perm"] = perm
        if name != "clustered20k":
            sort_cases[f"{name}_raw"] = raw_s
        else:
            sort_cases[f"{name}_rawsum"] = np.array([int(raw_s.astype(np.uint64).sum())], np.uint64)
    np.savez_compressed(os.path.join(OUT, "sort.npz"), **sort_cases)

    # ---- (3) LOD cuts ----
    raw_l = synth.clustered_cloud(4000, seed=6)
    split_l = R.raw_to_split(raw_l)
    c = split_l[:, 0:3].mean(axis=0, dtype=np.float32)  # any centring works; it is stored with the fixture
    split_l[:, 0:3] -= c
    oc = R.octree_build(split_l)
    lod = dict(raw=raw_l, split=split_l, oc_splats=oc["splats"], oc_bb16=oc["bb16"], oc_center3=oc["center3"],
               oc_half3=oc["half3"], oc_first_child=oc["first_child"], oc_n_children=oc["n_children"])
    lod_poses = poses([0, 10, 45])
    lod["pose_world"] = np.stack([p["world"] for p in lod_poses])
    lod["pose_viewproj"] = np.stack([p["viewproj"] for p in lod_poses])
    thrs = np.array([0.002, 0.02, 0.0002], np.float32)
    lod["oc_thr"] = thrs
    for pi, p in enumerate(lod_poses):
        lod[f"oc_area_p{pi}"] = np.array([R.bb_screen_area(oc["bb16"][i], p["world"], 45.0, ASPECT)
                                          for i in range(len(oc["bb16"]))], np.float32)
        for ti, thr in enumerate(thrs):
            lod[f"oc_cut_p{pi}_t{ti}"] = R.octree_get_indices(oc["handle"], p["world"], float(thr), 45.0, ASPECT)
    splats_l = R.split_to_splat(split_l)
    hc = R.hc_build(splats_l[:300])
    lod.update(hc_splats=hc["splats"], hc_child2=hc["child2"], hc_error=hc["error"], hc_roots=hc["roots"],
               hc_weight=R.splat_weight(hc["splats"]))
    depths = np.array([0, 1, 5, 50, 299, 1000], np.uint32)
    lod["hc_depths"] = depths
    for d in depths:
        lod[f"hc_cut_depth_{d}"] = R.hc_get_indices_depth(hc["handle"], int(d))
    mthr = np.array([1e-6, 1e-4, 1e-2, 1.0], np.float32)
    lod["hc_mthr"] = mthr
    for pi, p in enumerate(lod_poses):
        for ti, thr in enumerate(mthr):
            lod[f"hc_cut_metric_p{pi}_t{ti}"] = R.hc_get_indices(hc["handle"], p["world"], float(thr))
    g = R.gridhc_build(splats_l, (4, 4, 4))
    lod.update(g_splats=g["splats"], g_child2=g["child2"], g_error=g["error"], g_roots=g["roots"],
               g_cell_root_start=g["cell_root_start"], g_weight=R.splat_weight(g["splats"]))
    gdepths = np.array([0, 3, 40, 2000], np.uint32)
    lod["g_depths"] = gdepths
    for d in gdepths:
        lod[f"g_cut_depth_{d}"] = R.gridhc_get_indices_error(g["handle"], int(d))
    gthr = np.array([1e-5, 1e-3], np.float32)
    lod["g_mthr"] = gthr
    for pi, p in enumerate(lod_poses):
        for ti, thr in enumerate(gthr):
            lod[f"g_cut_metric_p{pi}_t{ti}"] = R.gridhc_get_indices(g["handle"], p["world"], float(thr))
    np.savez_compressed(os.path.join(OUT, "lod.npz"), **lod)
    for f in ("datamodel.npz", "camera_orbit.npz", "sort.npz", "lod.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
