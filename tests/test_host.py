"""CPU: libgsx's host-only entry points (gsx_host.cpp on host/gsx_*.hpp) against the
reference-generated fixtures, bit for bit.  No GPU is touched."""
import importlib

import numpy as np

from conftest import bits, golden


def test_convert_raw_matches_reference(pkg):
    scene = importlib.import_module("gaussian_splatting_amd.scene")
    g = golden("datamodel.npz")
    assert np.array_equal(bits(scene.raw_to_splats(pkg, g["raw"])), bits(g["splat"]))       # Splat.h:60-91


def test_convert_raw_centering_matches_oracle(pkg, oracle, synth):
    scene = importlib.import_module("gaussian_splatting_amd.scene")
    raw = synth.uniform_cloud(5000, seed=21)
    assert np.array_equal(bits(scene.raw_to_splats(pkg, raw, center=True)), bits(oracle.raw_to_splat(raw, center=True)))  # ResourceManager.cpp:52-62
    assert scene.raw_to_splats(pkg, raw[:0]).shape == (0, 20)


def test_load_splat_file(pkg, synth, tmp_path):
    scene = importlib.import_module("gaussian_splatting_amd.scene")
    raw = synth.uniform_cloud(777, seed=3)
    path = tmp_path / "s.splat"
    with open(path, "wb") as f:
        f.write(raw.tobytes() + b"\x01\x02\x03")     # trailing partial record is ignored (file.read fails)
    a = scene.load_splat_file(pkg, path, center=True)
    assert np.array_equal(bits(a), bits(scene.raw_to_splats(pkg, raw, center=True)))
    assert scene.load_splat_file(pkg, tmp_path / "missing.splat").shape == (0, 20)           # ResourceManager.cpp:38-40


def test_camera_orbit_matches_reference(pkg):
    cam = importlib.import_module("gaussian_splatting_amd.camera")
    g = golden("camera_orbit.npz")
    angle = float(g["angle"])
    for s in (0, 1, 2, 17, 60, 119):
        world = cam.orbit_pose(5.0, [((0.0, 1.0, 0.0), angle)] * s)
        assert np.array_equal(bits(world), bits(g["world"][s])), s                           # Node.cpp:10-27
        view, proj, pos, vp = cam.camera_matrices(world, 45.0, float(g["aspect"]), want_viewproj=True)
        assert np.array_equal(bits(view), bits(g["view"][s]))                                # Camera.h:27-29
        assert np.array_equal(bits(proj), bits(g["proj"][s]))                                # Camera.h:32-34
        assert np.array_equal(bits(vp), bits(g["viewproj"][s]))                              # BB.hpp:60-61
