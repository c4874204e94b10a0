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


# ---- the optional viewer front end (host/gsx_viewer.hpp, gaussian-splatting_amd/gsx_view): the reference's main loop fed by an
# event script, frames out as PPM files and a raw RGBA stream ----
VIEW_SCRIPT = """\
# a camera at rest, then a drag with the left button, a dolly, and the panel's sliders
dt 0.0166666
frame 2
move 300 200
button 0 1
move 340 215
frame
move 420 260
move 470 240
frame 2
button 0 0
move 10 10
scroll 3
frame
set splatSize 0.5
set cutOff 2.5
set fov 60
frame
set bayerSize 2
set orbitRate 2
button 0 1
move 60 40
scroll -20
frame 2
quit
frame 5
"""


def _script_frames(text):
    """What the script means, frame by frame: (events delivered so far as gsx_host_orbit_events rows with the rates the orbit camera
    holds at that moment, slider values) — rebuilt here independently of the C++ loop (main.cpp:209-236 order: poll, then refresh)."""
    frames, ev, dt = [], [], 1 / 60
    gui = dict(splatSize=1.0, cutOff=3.5, fov=45.0, bayerSize=1, bayerScale=1.0, orbitRate=1.0, scrollRate=1.0)
    for line in text.splitlines():
        w = line.split()
        if not w or w[0].startswith("#"):
            continue
        if w[0] == "dt":
            dt = float(w[1])
        elif w[0] == "move":
            ev.append([0.0, float(w[1]), float(w[2]), dt])
        elif w[0] == "button":
            ev.append([1.0, float(w[1]), float(w[2]), 0.0])
        elif w[0] == "scroll":
            ev.append([2.0, float(w[1]), 0.0, dt])
        elif w[0] == "set":
            gui[w[1]] = type(gui[w[1]])(float(w[2]))
        elif w[0] == "frame":
            for _ in range(int(w[1]) if len(w) > 1 else 1):
                frames.append((len(ev), dict(gui)))
        elif w[0] == "quit":
            break
    return frames, np.asarray(ev, np.float64).reshape(-1, 4)


def test_viewer_front_end(tmp_path, pkg, synth, oracle):
    cam = importlib.import_module("gaussian_splatting_amd.camera")
    exe = os.path.join(ROOT, "gaussian-splatting_amd", "gsx_view")
    assert os.path.exists(exe), "build it with make -C gaussian-splatting_amd/csrc (or __graft_entry__.build())"
    raw = synth.clustered_cloud(20000, seed=23)
    scene = tmp_path / "scene.splat"
    raw.tofile(scene)
    (tmp_path / "ev.txt").write_text(VIEW_SCRIPT)
    out = tmp_path / "frames"
    out.mkdir()
    r = subprocess.run([exe, str(scene), "--size", f"{W}x{H}", "--events", str(tmp_path / "ev.txt"), "--out", str(out), "--raw", str(tmp_path / "raw.rgba"),
                        "--uniforms", str(tmp_path / "uni.bin"), "--stats", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    frames, ev = _script_frames(VIEW_SCRIPT)
    assert len(frames) == 9 and f"{len(frames)} frames presented" in r.stderr          # (nothing after `quit` runs)
    assert "ms/frame (" in r.stderr and "FPS)" in r.stderr                              # the panel's statistics line, gui.cpp:130-131
    rawf = np.fromfile(tmp_path / "raw.rgba", np.uint8).reshape(len(frames), H, W, 4)
    rec = np.fromfile(tmp_path / "uni.bin", np.uint8).reshape(len(frames), 224 + 64)
    splats = oracle.raw_to_splat(raw, center=True)
    g = pkg.Gsx(0)
    try:
        g.upload(splats)
        seen = set()
        for k, (n_ev, gui) in enumerate(frames):
            ppm = open(out / f"frame_{k:05d}.ppm", "rb").read()
            head = f"P6\n{W} {H}\n255\n".encode()
            assert ppm.startswith(head) and np.array_equal(np.frombuffer(ppm[len(head):], np.uint8).reshape(H, W, 3), rawf[k][..., :3])
            # the pose: OrbitCamera's handlers over the events delivered so far.  The handlers read the rates the loop copied from the
            # panel at the end of the PREVIOUS frame (main.cpp:235-236): 0.3 / 5.0 (OrbitCamera.h:18-21) during the first poll only.
            u = pkg.Uniforms.from_buffer_copy(rec[k][:224].tobytes())
            world = rec[k][224:].view(np.float32)
            rates = [(0.3, 5.0)] + [(f[1]["orbitRate"], f[1]["scrollRate"]) for f in frames]    # rates[j]: in force while frame j polls
            bounds = [0] + [f[0] for f in frames]                                                 # frame j polls events bounds[j] .. bounds[j + 1]
            assert all(rates[j] == (1.0, 1.0) for j in range(k + 1) if bounds[j + 1] > bounds[j])  # (this script: one pair of rates for every event)
            assert np.array_equal(bits(world), bits(cam.orbit_events(ev[:n_ev], 5.0, 1.0, 1.0))), k
            view, proj, pos = cam.camera_matrices(world, u.fov if k == 0 else frames[k - 1][1]["fov"], W / H)   # UpdateScene: last frame's fov
            assert np.array_equal(bits(np.frombuffer(u.viewMatrix, np.float32)), bits(view)) and np.array_equal(bits(np.frombuffer(u.projectionMatrix, np.float32)), bits(proj)), k
            assert (u.splatSize, u.cutOff, u.fov, u.bayerSize) == (np.float32(gui["splatSize"]), np.float32(gui["cutOff"]), np.float32(gui["fov"]), gui["bayerSize"]), k
            # the frame: what the library draws for that uniform block and camera position
            g.sort(pos, download=False)
            assert np.array_equal(g.render(u, W, H), rawf[k]), k
            seen.add(rawf[k].tobytes())
        assert np.array_equal(rawf[0], rawf[1]) and len(seen) >= 6                      # a camera at rest repeats its frame; the drag, the dolly and the sliders change it
        assert rawf[..., :3].std() > 3
    finally:
        g.close()
    # pipelined presentation (every frame shown two loops late, both frame slots busy): the same frames and records in the same order
    r = subprocess.run([exe, str(scene), "--size", f"{W}x{H}", "--events", str(tmp_path / "ev.txt"), "--raw", str(tmp_path / "raw2.rgba"),
                        "--uniforms", str(tmp_path / "uni2.bin"), "--stats", "0", "--pipelined"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and f"{len(frames)} frames presented" in r.stderr, r.stdout + r.stderr
    assert open(tmp_path / "raw2.rgba", "rb").read() == open(tmp_path / "raw.rgba", "rb").read()
    assert open(tmp_path / "uni2.bin", "rb").read() == open(tmp_path / "uni.bin", "rb").read()
    r = subprocess.run([exe, str(scene), "--size", f"{W}x{H}", "--frames", "1", "--pipelined", "--raw", str(tmp_path / "one.rgba")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "1 frames presented" in r.stderr and np.array_equal(np.fromfile(tmp_path / "one.rgba", np.uint8), rawf[0].reshape(-1))
    # an LOD mesh through the same loop (the reference's choice, main.cpp:98, is GridHC), events from a pipe, no script file
    r = subprocess.run([exe, str(scene), "--mesh", "octree", "--size", f"{W}x{H}", "--events", "-", "--raw", str(tmp_path / "oct.rgba"), "--stats", "0"],
                       input="set min_screen_area 1\nframe\nmove 0 0\nbutton 0 1\nmove 30 0\nframe\n", capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "2 frames presented" in r.stderr, r.stdout + r.stderr
    octf = np.fromfile(tmp_path / "oct.rgba", np.uint8).reshape(2, H, W, 4)
    assert octf[..., :3].std() > 3 and not np.array_equal(octf[0], octf[1])
    # the two clustered meshes (SplatMeshHC keeps 500 splats, SplatMeshHC.hpp:72-75; GridHC is the reference's choice): the `depth` slider
    # moves the view-independent cut, so the frame changes with it while the camera rests
    small = tmp_path / "small.splat"
    synth.uniform_cloud(3000, seed=17, half_extent=1.5).tofile(small)
    for mesh in ("hc", "gridhc"):
        r = subprocess.run([exe, str(small), "--mesh", mesh, "--size", f"{W}x{H}", "--events", "-", "--raw", str(tmp_path / f"{mesh}.rgba"), "--stats", "1"],
                           input="set depth 2\nframe 2\nset depth 400\nframe\n", capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "3 frames presented" in r.stderr, r.stdout + r.stderr
        f = np.fromfile(tmp_path / f"{mesh}.rgba", np.uint8).reshape(3, H, W, 4)
        assert np.array_equal(f[0], f[1]) and not np.array_equal(f[1], f[2]) and f[..., :3].std() > 3, mesh
        drawn = [int(line.split()[-3]) for line in r.stderr.splitlines() if line.endswith("splats drawn")]
        assert len(drawn) == 2 and drawn[0] < drawn[1], (mesh, drawn)      # (no rate yet after the first frame) a deeper cut draws more splats
    # errors are loud: a slider that does not exist, a line that is no event
    # the example script that ships with the program
    example = os.path.join(ROOT, "gaussian-splatting_amd", "viewer", "orbit_drag.txt")
    r = subprocess.run([exe, str(small), "--mesh", "gridhc", "--size", f"{W}x{H}", "--events", example, "--stats", "0", "--every", "10", "--out", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "43 frames presented" in r.stderr, r.stdout + r.stderr
    assert os.path.exists(out / "frame_00040.ppm") and not os.path.exists(out / "frame_00041.ppm")      # --every 10
    # (bayerSize 3: the slider of gui.cpp:60 allows it, the shader's Bayer matrix does not exist for it; gsx_render rejects the block)
    for bad, what in (("set nosuch 1\nframe\n", "viewer script"), ("wiggle 3\n", "viewer script"), ("set bayerSize 3\nframe\n", "bayerSize")):
        r = subprocess.run([exe, str(scene), "--size", f"{W}x{H}", "--events", "-"], input=bad, capture_output=True, text=True, timeout=300)
        assert r.returncode == 1 and what in r.stderr, (bad, r.stderr)
