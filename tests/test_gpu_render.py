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


def test_render_repeated_frames_on_one_context(gsx, pkg, oracle, synth):
    # a viewer renders frame after frame on one context: nothing (tile state, accumulators, LDS leftovers)
    # may leak from one frame into the next
    w, h = 256, 192
    splats = scene(oracle, synth, 3000, seed=11)
    gsx.upload(splats)
    first = {}
    for rnd in range(3):
        for steps in (0, 4):
            view, proj, cam = camera(oracle, w, h, steps)
            perm = gsx.sort(cam)
            img = gsx.render(pkg.make_uniforms(proj, view), w, h)
            if steps not in first:
                check_image(img, oracle.render(splats, perm, orc.make_uniforms(proj, view), w, h))
                first[steps] = img.copy()
            assert np.array_equal(img, first[steps]), f"frame changed on round {rnd}"


def dense_scene(oracle, synth, n, seed):
    # large opaque-ish splats: every tile saturates after a few dozen splats, so tiles retire and the
    # binner's occlusion culling is what the frame depends on
    return oracle.raw_to_splat(synth.clustered_cloud(n, seed=seed, n_clusters=6, sigma=0.5, frac=0.8, centre_extent=1.0, scale_mul=5.0))


@pytest.mark.parametrize("opts", [
    {},                                                            # default schedule (one iteration at this size)
    {"chunk_first": 256, "chunk_growth": 2},                       # 8 scheduled iterations
    {"chunk_first": 512, "chunk_growth": 2, "pair_budget": 20000},  # budget-cut iterations + continuation on the host
    {"chunk_first": 1024, "pair_budget": 64},                      # nearest block alone exceeds the budget: buffers grow
])
def test_render_iterations_and_pair_budget(gsx_fresh, pkg, oracle, synth, opts):
    gsx = gsx_fresh
    w, h = 240, 176
    splats = dense_scene(oracle, synth, 20000, seed=5)
    view, proj, cam = camera(oracle, w, h, steps=2)
    for k, v in opts.items():
        gsx.set_option(k, v)
    gsx.upload(splats)
    perm = gsx.sort(cam)
    img = gsx.render(pkg.make_uniforms(proj, view), w, h)
    t = gsx.timings()
    ref = oracle.render(splats, perm, orc.make_uniforms(proj, view), w, h)
    check_image(img, ref)
    assert ref[..., :3].std() > 5
    if "chunk_first" in opts and "pair_budget" not in opts:
        assert t["render_iters"] >= 2                              # later iterations skip what is hidden
    # same frame again, and a tile band of it, on the same context
    assert np.array_equal(gsx.render(pkg.make_uniforms(proj, view), w, h), img)
    band = np.zeros_like(img)
    gsx.render(pkg.make_uniforms(proj, view), w, h, 32, 112, out=band)
    # iterations cut by the pair budget end at other ranks for a band than for the whole frame; pixels freeze at their own
    # first splat with T < 2^-11 (gsx_raster.hip), so the picture does not depend on where iterations end
    assert np.array_equal(band[32:112], img[32:112])


def test_render_occlusion_culling_skips_hidden_pairs(gsx_fresh, pkg, oracle, synth):
    gsx = gsx_fresh
    w, h = 240, 176
    # many small opaque-ish splats in a tight cloud: the centre tiles saturate within the nearest few hundred
    splats = oracle.raw_to_splat(synth.clustered_cloud(60000, seed=8, n_clusters=3, sigma=0.3, frac=0.95, centre_extent=0.5, scale_mul=4.0))
    view, proj, cam = camera(oracle, w, h, steps=2, distance=2.5)
    gsx.upload(splats)
    gsx.sort(cam)
    u = pkg.make_uniforms(proj, view)
    gsx.set_option("chunk_first", 1 << 30)
    one = gsx.render(u, w, h); p_one = gsx.timings()["tile_pairs"]; i_one = gsx.timings()["render_iters"]
    gsx.set_option("chunk_first", 512); gsx.set_option("chunk_growth", 2)
    many = gsx.render(u, w, h); p_many = gsx.timings()["tile_pairs"]; i_many = gsx.timings()["render_iters"]
    assert p_many < 0.6 * p_one, (p_many, p_one, i_many, i_one)
    assert np.array_equal(one, many)                               # the schedule decides what is skipped, never a pixel


def test_set_option_rejects_unknown(gsx_fresh, pkg):
    gsx = gsx_fresh
    with pytest.raises(pkg.GsxError):
        gsx.set_option("no_such_knob", 1)
    with pytest.raises(pkg.GsxError):
        gsx.set_option("chunk_growth", 1)


def test_large_render_properties(gsx_fresh, pkg, oracle, synth):
    """cfg-2 size (6M splats, 1920x1080): the image oracle would take minutes, so check size-independent
    properties of the frame loop instead."""
    gsx = gsx_fresh
    n, w, h = 6_000_000, 1920, 1080
    splats = oracle.raw_to_splat(synth.clustered_cloud(n, seed=6))
    view, proj, cam = camera(oracle, w, h)
    u = pkg.make_uniforms(proj, view)
    gsx.upload(splats)
    perm = gsx.sort(cam)
    gsx.set_option("no_history", 1)
    img = gsx.render(u, w, h)
    t = gsx.timings()
    assert np.all(img[..., 3] == 255) and img[..., :3].std() > 10
    assert 2 <= t["render_iters"] <= 8 and t["tile_pairs"] < 40_000_000      # occlusion culling at work (one shot: 76 M pairs)
    # (1) determinism
    assert np.array_equal(gsx.render(u, w, h), img)
    # (2) tile bands (the multi-GPU sharding unit) compose byte-exactly when they share the rank schedule
    out = np.zeros_like(img)
    for y0, y1 in ((0, 272), (272, 544), (544, 816), (816, 1080)):
        gsx.render(u, w, h, y0, y1, out=out)
    assert np.array_equal(out, img)
    # (3) one-shot binning (no occlusion culling inside the iteration; cut only by the pair budget) gives the same
    #     picture byte for byte
    gsx.set_option("chunk_first", 1 << 30)
    one = gsx.render(u, w, h)
    t1 = gsx.timings()
    assert t1["tile_pairs"] > 2 * t["tile_pairs"]
    assert np.array_equal(one, img)
    gsx.set_option("chunk_first", 0)
    # (4) everything behind the nearest third of the splats is hidden in this scene: drawing only those gives the same picture
    gsx.select_list(np.sort(perm[-2_000_000:]))
    gsx.sort(cam)
    near = gsx.render(u, w, h)
    assert np.array_equal(near, img)
    gsx.select_all()
    # (5) with the history-based schedule (default) frames are byte-identical to the history-free frame
    gsx.set_option("no_history", 0)
    gsx.sort(cam)
    for _ in range(3):
        again = gsx.render(u, w, h)
        assert np.array_equal(again, img)
    # (6) a steady camera INSIDE the cloud (dense foreground: small first window, and the schedule learns to take
    #     everything behind it in one iteration): frames under the adapting schedule agree with the history-free frame
    view2, proj2, cam2 = camera(oracle, w, h, steps=1, distance=2.0)
    u2 = pkg.make_uniforms(proj2, view2)
    gsx.set_option("no_history", 1)
    gsx.sort(cam2)
    inside = gsx.render(u2, w, h)
    assert inside[..., :3].std() > 10
    gsx.set_option("no_history", 0)
    for _ in range(5):
        again = gsx.render(u2, w, h)
        assert np.array_equal(again, inside)


def test_large_overlap_4k_bands(gsx_fresh, pkg, oracle, synth):
    """cfg 5: 6M-splat overlap-stress scene at 3840x2160 in 8 horizontal bands (one per GPU of a node)."""
    gsx = gsx_fresh
    n, w, h = 6_000_000, 3840, 2160
    splats = oracle.raw_to_splat(synth.overlap_stress_cloud(n))
    view, proj, cam = camera(oracle, w, h)
    u = pkg.make_uniforms(proj, view)
    gsx.upload(splats)
    gsx.sort(cam)
    gsx.set_option("no_history", 1)
    img = gsx.render(u, w, h)
    assert img[..., :3].std() > 10
    aligned = np.zeros_like(img)
    for b in range(8):                                   # tile-aligned bands: 7 x 272 rows + 256
        gsx.render(u, w, h, b * 272, min((b + 1) * 272, h), out=aligned)
    # (the pair budget cuts this scene's iterations, and it cuts a band elsewhere than the whole frame: no matter)
    assert np.array_equal(aligned, img)
    ragged = np.zeros_like(img)
    for b in range(8):                                   # 270-row bands as in SURVEY 8(d): boundaries inside tiles
        gsx.render(u, w, h, b * 270, (b + 1) * 270, out=ragged)
    assert np.array_equal(ragged, img)
    gsx.set_option("no_history", 0)                      # the default, history-driven schedule: same bytes
    gsx.sort(cam)
    for _ in range(2):
        assert np.array_equal(gsx.render(u, w, h), img)


def test_20m_scene_properties(gsx_fresh, pkg, oracle, synth):
    """cfg-4 size (20M splats): sort and frame loop at the largest configured scene, by properties."""
    gsx = gsx_fresh
    n, w, h = 20_000_000, 1920, 1080
    splats = oracle.raw_to_splat(synth.clustered_cloud(n, seed=20))
    gsx.upload(splats)
    for steps in (0, 3):
        view, proj, cam = camera(oracle, w, h, steps)
        perm = gsx.sort(cam)
        k = oracle.distance_keys(splats, cam)[perm]
        assert np.all(k[:-1] >= k[1:])                                   # sortedness
        same = k[1:] == k[:-1]
        assert np.all(perm[1:][same] > perm[:-1][same])                  # ties by ascending index
        assert np.array_equal(np.bincount(perm, minlength=n), np.ones(n, np.int64))   # a permutation
        u = pkg.make_uniforms(proj, view)
        img = gsx.render(u, w, h)
        assert np.all(img[..., 3] == 255) and img[..., :3].std() > 10
        assert np.array_equal(gsx.render(u, w, h), img)
    del splats


def _read_device(ptr, nbytes):
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    out = np.empty(nbytes, np.uint8)
    rc = hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(nbytes), C.c_int(2))   # device -> host
    assert rc == 0
    return out


def test_frames_queued_back_to_back(gsx_fresh, pkg, oracle, synth):
    # a viewer queues frame k+1's sort while frame k is still running: the sort writes the other permutation buffer, and
    # continuation iterations of frame k (forced here by a small pair budget) are queued AFTER that sort and must still
    # draw frame k's permutation
    gsx = gsx_fresh
    w, h = 240, 176
    splats = dense_scene(oracle, synth, 20000, seed=5)
    for k, v in {"chunk_first": 512, "chunk_growth": 2, "pair_budget": 20000, "no_history": 1}.items():
        gsx.set_option(k, v)
    gsx.upload(splats)
    cams = [camera(oracle, w, h, steps=s) for s in (0, 3, 6)]
    unis = [pkg.make_uniforms(proj, view) for view, proj, _ in cams]
    ref, perms = [], []
    for (view, proj, cam), u in zip(cams, unis):          # one frame at a time, each read back before the next
        perms.append(gsx.sort(cam))
        ref.append(gsx.render(u, w, h).copy())
    check_image(ref[1], oracle.render(splats, perms[1], orc.make_uniforms(cams[1][1], cams[1][0]), w, h))
    assert not np.array_equal(ref[0], ref[1])
    gsx.set_profiling(2)
    gsx.scatter_samples(reset=True)
    gsx.sort(cams[0][2], download=False)
    for k in range(3):
        gsx.render(unis[k], w, h, download=False)
        if k + 1 < 3:
            gsx.sort(cams[k + 1][2], download=False)       # queued behind frame k without waiting for it
        gsx.sync()                                         # finishes frame k (its extra iterations run after that sort)
        assert gsx.timings()["render_iters"] > 4
        ptr, fw, fh = gsx.framebuffer_dev()
        assert (fw, fh) == (w, h)
        img = _read_device(ptr, w * h * 4).reshape(h, w, 4)
        assert np.array_equal(img, ref[k]), f"frame {k} changed when the next sort was queued early"
    # two sorts in a row without a frame between them, then a frame of the later one
    gsx.sort(cams[0][2], download=False)
    gsx.sort(cams[1][2], download=False)
    assert np.array_equal(gsx.render(unis[1], w, h), ref[1])
    assert np.array_equal(gsx.sort(cams[2][2]), perms[2])
    means, launches = gsx.scatter_samples()
    whole_ms, whole_n = gsx.sort_samples()
    # six sorts were sampled: the rotation goes pass 0, 1, ..., then keys + sort as a whole
    assert sum(launches.values()) + whole_n == 6 and whole_n >= 1 and all(v > 0.0 for v in means.values())
    passes = gsx.timings()["sort_passes"]
    assert set(launches) <= set(range(min(passes, 3))) and len(launches) >= 2
    assert whole_ms > max(means.values())                    # the whole sort takes longer than any one of its launches


def test_render_into_caller_owned_target(gsx_fresh, pkg, oracle, synth):
    # gsx_set_framebuffer_dev: frames go straight into caller-owned device memory (the slots of the multi-GPU view
    # batches); byte-identical to the library's own framebuffer, nothing written past width*height*4
    import torch
    gsx = gsx_fresh
    w, h = 256, 192
    nbytes = w * h * 4
    splats = scene(oracle, synth, 3000, seed=2)
    view, proj, cam = camera(oracle, w, h, steps=1)
    u = pkg.make_uniforms(proj, view)
    gsx.upload(splats)
    gsx.sort(cam)
    img = gsx.render(u, w, h).copy()
    buf = torch.full((2 * nbytes + 256,), 7, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    for slot in (1, 0):                                       # two slots of one batch buffer, like bench.py at N > 1
        gsx.set_framebuffer_dev(buf.data_ptr() + slot * nbytes, nbytes)
        gsx.sort(cam, download=False)
        gsx.render(u, w, h, download=False)
    gsx.set_framebuffer_dev(None)                             # completes the queued frame into its slot
    gsx.sync()
    host = buf.cpu().numpy()
    for slot in (0, 1):
        assert np.array_equal(host[slot * nbytes:(slot + 1) * nbytes].reshape(h, w, 4), img)
    assert np.all(host[2 * nbytes:] == 7)
    # the library's stream is a usable torch stream: work queued on it is ordered behind the frame
    ext = torch.cuda.ExternalStream(gsx.stream(), device=torch.device("cuda", 0))
    gsx.set_framebuffer_dev(buf.data_ptr(), nbytes)
    gsx.render(u, w, h, download=False)
    gsx.set_framebuffer_dev(None)
    with torch.cuda.stream(ext):
        copy = buf[:nbytes].clone()
    ext.synchronize()
    assert np.array_equal(copy.cpu().numpy().reshape(h, w, 4), img)
    # a target that is too small is rejected, and the context keeps working on its own framebuffer afterwards
    gsx.set_framebuffer_dev(buf.data_ptr(), nbytes - 4)
    with pytest.raises(pkg.GsxError):
        gsx.render(u, w, h, download=False)
    gsx.set_framebuffer_dev(None)
    assert np.array_equal(gsx.render(u, w, h), img)
    # band into an external target leaves the other rows alone
    buf.fill_(9)
    torch.cuda.synchronize()
    gsx.set_framebuffer_dev(buf.data_ptr(), nbytes)
    gsx.render(u, w, h, 32, 112, download=False)
    gsx.set_framebuffer_dev(None)
    gsx.sync()
    band = buf[:nbytes].cpu().numpy().reshape(h, w, 4)
    assert np.array_equal(band[32:112], img[32:112]) and np.all(band[:32] == 9) and np.all(band[112:] == 9)
