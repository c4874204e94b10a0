"""GPU: the drop-in C++ classes (host/gsx_facade.hpp: Renderer, SplatMesh, SplatMeshOctree, SplatMeshHC,
SplatMeshGridHC) driven like the reference's Application by tests/cpp/facade_test, checked against the oracle."""
import importlib
import os
import subprocess

import numpy as np
import pytest

import orc
from conftest import ROOT, bits

pytestmark = pytest.mark.gpu
W, H = 224, 160


def _load(out, name):
    splats = np.fromfile(os.path.join(out, name + ".splats"), np.float32).reshape(-1, 20)
    idx = np.fromfile(os.path.join(out, name + ".indices"), np.uint32)
    img = np.fromfile(os.path.join(out, name + ".rgba"), np.uint8).reshape(H, W, 4)
    u = orc.Uniforms.from_buffer_copy(open(os.path.join(out, name + ".uniforms"), "rb").read())
    world = np.fromfile(os.path.join(out, name + ".world"), np.float32)
    return splats, idx, img, u, world


@pytest.fixture(scope="module")
def run(tmp_path_factory, pkg, synth):
    exe = os.path.join(ROOT, "tests", "cpp", "facade_test")
    assert os.path.exists(exe), "build it with make -C gaussian-splatting_amd/csrc (or __graft_entry__.build())"
    out = str(tmp_path_factory.mktemp("facade"))
    raw = synth.uniform_cloud(3000, seed=17, half_extent=1.5)
    scene = os.path.join(out, "scene.splat")
    raw.tofile(scene)
    r = subprocess.run([exe, scene, out, str(W), str(H)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return out, raw


def _check_image(img, ref):
    err = np.abs(img.astype(np.int32) - ref.astype(np.int32))
    assert err.max() <= 2
    mse = np.mean((img[..., :3].astype(np.float64) - ref[..., :3].astype(np.float64)) ** 2)
    assert mse == 0 or 10 * np.log10(255.0 ** 2 / mse) >= 45.0


def test_base_class(run, oracle):
    out, raw = run
    splats, idx, img, u, world = _load(out, "base")
    assert np.array_equal(bits(splats), bits(oracle.raw_to_splat(raw, center=True)))      # loadData(path, center=true)
    cam = world[12:15]
    assert np.array_equal(idx, oracle.sort_contract(splats, cam))                         # render(): sortSplats(indices, cameraPos)
    _check_image(img, oracle.render(splats, idx, u, W, H))
    assert img[..., :3].std() > 3


def test_octree_class(run, oracle, pkg):
    out, raw = run
    lod = importlib.import_module("gaussian_splatting_amd.lod")
    splats, idx, img, u, world = _load(out, "octree")
    t = lod.build_octree(raw, center=True)                                                # host builder, pinned in test_host.py
    assert np.array_equal(bits(splats), bits(t.splats))
    vp = oracle.mul(np.frombuffer(u.proj, np.float32), np.frombuffer(u.view, np.float32))
    cut = oracle.octree_cut(t.center3, t.half3, t.first_child, t.n_children, vp, np.float32(1.5) * np.float32(0.01))
    assert 0 < len(cut) < len(splats)
    assert np.array_equal(idx, oracle.sort_contract(splats, world[12:15], cut))
    _check_image(img, oracle.render(splats, idx, u, W, H))


def test_hc_class(run, oracle, pkg):
    out, raw = run
    lod = importlib.import_module("gaussian_splatting_amd.lod")
    splats, idx, img, u, world = _load(out, "hc")
    base = oracle.raw_to_splat(raw, center=True)[:500]                                    # SplatMeshHC.hpp:72-75 keeps 500
    t = lod.build_hc(base)
    assert np.array_equal(bits(splats), bits(t.splats))
    cut = oracle.hc_cut_depth(t.roots, t.child2, 40)
    assert np.array_equal(idx, oracle.sort_contract(splats, world[12:15], np.sort(cut)))
    _check_image(img, oracle.render(splats, idx, u, W, H))


def test_gridhc_class(run, oracle, pkg):
    out, raw = run
    lod = importlib.import_module("gaussian_splatting_amd.lod")
    splats, idx, img, u, world = _load(out, "gridhc")
    base = oracle.raw_to_splat(raw, center=True)
    base = base[base[:, 12] > 0.0]                                                        # SplatMeshGridHC.cpp:37-43 keeps x > 0
    t = lod.build_gridhc(base, (20, 20, 20))
    assert np.array_equal(bits(splats), bits(t.splats))
    crs = t.cell_root_start
    cut = np.concatenate([oracle.hc_cut_depth(np.ascontiguousarray(t.roots[crs[c]:crs[c + 1]]), t.child2, 3)
                          for c in range(len(crs) - 1)])
    assert np.array_equal(idx, oracle.sort_contract(splats, world[12:15], np.sort(cut)))
    _check_image(img, oracle.render(splats, idx, u, W, H))


def test_missing_file_renders_clear_colour(run):
    out, _ = run
    splats, idx, img, u, world = _load(out, "missing")
    assert len(splats) == 0 and len(idx) == 0
    assert np.all(img[..., :3] == 13) and np.all(img[..., 3] == 255)                      # (0.05,0.05,0.05,1)
