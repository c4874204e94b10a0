"""GPU parity: projection + binning + tile rasterizer (through gsx_render) against the
CPU restatement of the WGSL program (oracle/orc_raster.cpp).

Tolerance (BASELINE.json north_star): PSNR >= 45 dB and max abs error <= 2/255.
The image oracle is our restatement; the reference's WebGPU render is unpinned."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


def check_image(img, ref):
    err = np.abs(img.astype(np.int32) - ref.astype(np.int32))
    assert err.max() <= 2, f"max abs error {err.max()}/255"
    assert psnr(img[..., :3], ref[..., :3]) >= 45.0, f"PSNR {psnr(img[..., :3], ref[..., :3]):.2f} dB"
    assert np.all(img[..., 3] == 255)


def scene(oracle, synth, n, seed, kind="uniform"):
    raw = synth.uniform_cloud(n, seed=seed, half_extent=1.5) if kind == "uniform" else synth.clustered_cloud(n, seed=seed)
    return oracle.raw_to_splat(raw)


def camera(oracle, w, h, steps=0, distance=5.0):
    world, _ = oracle.orbit_world(distance, (0.0, 1.0, 0.0), 0.35, steps)
    view, proj, vp, cam = oracle.camera(world, 45.0, w / h)
    return view, proj, cam


@pytest.mark.parametrize("w,h,n,steps", [(256, 192, 3000, 0), (320, 200, 12000, 3), (130, 70, 800, 7)])
def test_render_matches_oracle(gsx, pkg, oracle, synth, w, h, n, steps):
    splats = scene(oracle, synth, n, seed=n % 13 + 1)
    view, proj, cam = camera(oracle, w, h, steps)
    gsx.upload(splats)
    perm = gsx.sort(cam)
    img = gsx.render(pkg.make_uniforms(proj, view), w, h)
    ref = oracle.render(splats, perm, orc.make_uniforms(proj, view), w, h)
    assert ref[..., :3].std() > 5          # the scene actually covers the frame
    check_image(img, ref)


def test_render_camera_inside_cloud_and_clipping(gsx, pkg, oracle, synth):
    # camera inside the cloud: splats behind the camera and in front of the (WebGPU 0..w) near plane are clipped
    w, h = 200, 160
    splats = scene(oracle, synth, 6000, seed=4)
    view, proj, cam = camera(oracle, w, h, steps=2, distance=0.6)
    gsx.upload(splats)
    perm = gsx.sort(cam)
    img = gsx.render(pkg.make_uniforms(proj, view), w, h)
    check_image(img, oracle.render(splats, perm, orc.make_uniforms(proj, view), w, h))


def test_render_model_matrix_and_params(gsx, pkg, oracle, synth):
    w, h = 192, 128
    splats = scene(oracle, synth, 4000, seed=9)
    view, proj, cam = camera(oracle, w, h, steps=1)
    model = np.eye(4, dtype=np.float32)
    model[:3, :3] *= 0.7
    model[3, :3] = [0.2, -0.1, 0.3]          # row-major numpy == column-major translation column
    model = model.reshape(-1)
    gsx.upload(splats)
    perm = gsx.sort(cam)
    for kw in (dict(splatSize=0.6, cutOff=3.5), dict(splatSize=1.0, cutOff=2.0)):
        img = gsx.render(pkg.make_uniforms(proj, view, model, **kw), w, h)
        check_image(img, oracle.render(splats, perm, orc.make_uniforms(proj, view, model, **kw), w, h))


def test_render_empty_and_lod_selection(gsx, pkg, oracle, synth):
    w, h = 160, 96
    splats = scene(oracle, synth, 5000, seed=3)
    view, proj, cam = camera(oracle, w, h)
    gsx.upload(splats)
    gsx.select_list(np.zeros(0, np.uint32))
    gsx.sort(cam)
    img = gsx.render(pkg.make_uniforms(proj, view), w, h)
    assert np.all(img[..., :3] == 13) and np.all(img[..., 3] == 255)      # clear colour 0.05 -> 13
    sub = np.arange(0, 5000, 3, dtype=np.uint32)
    gsx.select_list(sub)
    perm = gsx.sort(cam)
    img = gsx.render(pkg.make_uniforms(proj, view), w, h)
    check_image(img, oracle.render(splats, perm, orc.make_uniforms(proj, view), w, h))
    gsx.select_all()


def test_render_tile_bands_compose(gsx, pkg, oracle, synth):
    # screen-space tile bands (multi-GPU sharding unit): bands rendered separately == whole frame
    w, h = 224, 150
    splats = scene(oracle, synth, 5000, seed=6)
    view, proj, cam = camera(oracle, w, h, steps=5)
    u = pkg.make_uniforms(proj, view)
    gsx.upload(splats)
    gsx.sort(cam)
    whole = gsx.render(u, w, h)
    out = np.zeros((h, w, 4), np.uint8)
    for y0, y1 in ((0, 48), (48, 96), (96, 150)):
        gsx.render(u, w, h, y0, y1, out=out)
    assert np.array_equal(out, whole)


@pytest.mark.parametrize("bayer,scale", [(10, 0.3), (2, 1.0), (4, 1.0)])
def test_render_bayer_and_threshold_modes(gsx, pkg, oracle, synth, bayer, scale):
    w, h = 160, 120
    splats = scene(oracle, synth, 1500, seed=2)
    view, proj, cam = camera(oracle, w, h)
    gsx.upload(splats)
    perm = gsx.sort(cam)
    img = gsx.render(pkg.make_uniforms(proj, view, bayerSize=bayer, bayerScale=scale), w, h)
    ref = oracle.render(splats, perm, orc.make_uniforms(proj, view, bayerSize=bayer, bayerScale=scale), w, h)
    # hard 0/1 alpha: a pixel whose alpha sits on the threshold may flip; require 99.5 % exact pixels
    same = np.all(img == ref, axis=-1).mean()
    assert same >= 0.995, same


def test_render_rejects_bad_arguments(gsx, pkg, oracle, synth):
    splats = scene(oracle, synth, 100, seed=1)
    view, proj, cam = camera(oracle, 64, 64)
    gsx.upload(splats)
    with pytest.raises(pkg.GsxError) as e:
        gsx.render(pkg.make_uniforms(proj, view), 64, 64)        # no sort yet
    assert e.value.code == pkg.GSX_E_STATE
    gsx.sort(cam)
    with pytest.raises(pkg.GsxError):
        gsx.render(pkg.make_uniforms(proj, view), 64, 64, 10, 5)
    with pytest.raises(pkg.GsxError) as e:
        gsx.render(pkg.make_uniforms(proj, view, bayerSize=3), 64, 64)
    assert e.value.code == pkg.GSX_E_UNSUPPORTED
