"""GPU parity: projection + binning + tile rasterizer (through gsx_render) against the
CPU restatement of the WGSL program (oracle/orc_raster.cpp).

Tolerance (BASELINE.json north_star): PSNR >= 45 dB and max abs error <= 2/255.
The image oracle is our restatement; the reference's WebGPU render is unpinned."""
import numpy as np
import pytest

import known_answers as ka
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


def test_band_cull_before_the_sort(gsx_fresh, pkg, oracle, synth):
    """Tile-band sharding with the cull BEFORE the sort: gsx_select_band keeps the splats whose quad can reach the
    band's rows; the band's permutation is the full permutation with the culled entries removed, and the bands rendered from
    their own culled selections compose to the bytes of the whole frame."""
    gsx = gsx_fresh
    w, h, n = 640, 368, 150_000
    splats = oracle.raw_to_splat(synth.overlap_stress_cloud(n, seed=4))
    view, proj, cam = camera(oracle, w, h, steps=2)
    u = pkg.make_uniforms(proj, view)
    gsx.upload(splats)
    full = gsx.sort(cam)
    whole = gsx.render(u, w, h)
    assert whole[..., :3].std() > 5
    out = np.zeros_like(whole)
    kept = []
    for y0, y1 in ((0, 46), (46, 92), (92, 138), (138, 184), (184, 230), (230, 276), (276, 322), (322, 368)):
        for want in (True, False):
            m = gsx.select_band(u, w, h, y0, y1, want_count=want)
            ids = gsx.selection()
            assert np.all(ids[1:] > ids[:-1]) and (not want or m == len(ids))
            perm = gsx.sort(cam)
            assert np.array_equal(perm, full[np.isin(full, ids)])          # the full order with the culled entries removed
        gsx.render(u, w, h, y0, y1, out=out)
        kept.append(len(ids))
    assert np.array_equal(out, whole)
    assert min(kept) < 0.6 * n and sum(kept) < 4 * n, kept                  # the cull does remove most of a band's splats
    # back to everything: the whole frame again, same bytes
    gsx.select_all()
    assert np.array_equal(gsx.sort(cam), full)
    assert np.array_equal(gsx.render(u, w, h), whole)


@pytest.mark.parametrize("bayer,scale", [(10, 0.3), (2, 1.0), (4, 1.0)])
def test_render_bayer_and_threshold_modes(gsx, pkg, oracle, synth, bayer, scale):
    w, h = 160, 120
    splats = scene(oracle, synth, 1500, seed=2)
    view, proj, cam = camera(oracle, w, h)
    gsx.upload(splats)
    perm = gsx.sort(cam)
    img = gsx.render(pkg.make_uniforms(proj, view, bayerSize=bayer, bayerScale=scale), w, h)
    ref = oracle.render(splats, perm, orc.make_uniforms(proj, view, bayerSize=bayer, bayerScale=scale), w, h)
    # hard 0/1 alpha: the only pixels that may differ are those where some fragment's alpha sits ON the threshold, i.e. pixels
    # whose ORACLE value changes when bayerScale moves by 2e-4 relative (the two programs round alpha differently in the
    # last bits); every other pixel must agree exactly
    stable = np.ones((h, w), bool)
    for eps in (-2e-4, 2e-4):
        moved = oracle.render(splats, perm, orc.make_uniforms(proj, view, bayerSize=bayer, bayerScale=scale * (1.0 + eps)), w, h)
        stable &= np.all(moved == ref, axis=-1)
    assert stable.mean() >= 0.97, stable.mean()
    assert np.array_equal(img[stable], ref[stable]), int(np.any(img != ref, axis=-1)[stable].sum())
    assert np.all(img == ref, axis=-1).mean() >= 0.995


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


def _ka_render(gsx, pkg, splats, order, **kw):
    """Draw `splats` in exactly `order` (far -> near) through the C ABI: the permutation comes from gsx_sort, so the scenes
    below are laid out with distinct distances and the sort's result is asserted to BE that order."""
    view, proj = ka.matrices()
    gsx.upload(np.ascontiguousarray(splats))
    perm = gsx.sort(np.array([0.0, 0.0, 5.0], np.float32))
    assert np.array_equal(perm, np.asarray(order, np.uint32))
    return gsx.render(pkg.make_uniforms(proj, view, **kw), ka.W, ka.H)


def _ka_check(img, want):
    """GPU frame against a closed-form frame: within 1/255 everywhere (the GPU composites front to back, the closed form
    back to front), byte-exact away from 8-bit rounding boundaries."""
    got = img[..., :3].astype(np.int32)
    exp = ka.to_rgba8(want).astype(np.int32)
    assert np.abs(got - exp).max() <= 1
    ok = ~ka.near_rounding_boundary(want, margin=0.2)
    assert np.array_equal(got[ok], exp[ok])
    assert np.all(img[..., 3] == 255)


def test_known_answers_isotropic_and_blend_order(gsx, pkg):
    # tests/known_answers.py: hand-derived pixels (the same vectors pin the oracle in tests/test_oracle_known_answers.py)
    s = ka.splat((0, 0, 0), np.eye(3) * 0.01, (0.9, 0.4, 0.1, 0.8))[None]
    img = _ka_render(gsx, pkg, s, [0])
    assert tuple(img[50, 50]) == (176, 80, 22, 255) and tuple(img[50, 43]) == (14, 13, 13, 255) and tuple(img[52, 49]) == (90, 44, 17, 255)
    assert tuple(img[50, 42]) == (13, 13, 13, 255)                          # one pixel past the cutOff edge: clear colour
    _ka_check(img, ka.expected_image([((0, 0, 0), 0.1, 0.1, 0.0, (0.9, 0.4, 0.1, 0.8))]))
    far = ka.splat((0, 0, 0), np.eye(3) * 0.01, (0.1, 0.8, 0.3, 0.7))
    near = ka.splat((0, 0, 1), np.eye(3) * 0.01, (0.9, 0.2, 0.6, 0.5))
    img = _ka_render(gsx, pkg, np.stack([far, near]), [0, 1])               # camera at z = 5: index 0 is farther, drawn first
    assert tuple(img[50, 50]) == (121, 96, 102, 255)
    img = _ka_render(gsx, pkg, np.stack([near, far]), [1, 0])               # same scene, splats stored the other way round
    assert tuple(img[50, 50]) == (121, 96, 102, 255)
    _ka_check(img, ka.expected_image([((0, 0, 0), 0.1, 0.1, 0.0, (0.1, 0.8, 0.3, 0.7)), ((0, 0, 1), 0.1, 0.1, 0.0, (0.9, 0.2, 0.6, 0.5))]))
    # the colours swapped between the depths: what the OTHER draw order of the first scene would give
    far2 = ka.splat((0, 0, 0), np.eye(3) * 0.01, (0.9, 0.2, 0.6, 0.5))
    near2 = ka.splat((0, 0, 1), np.eye(3) * 0.01, (0.1, 0.8, 0.3, 0.7))
    img = _ka_render(gsx, pkg, np.stack([far2, near2]), [0, 1])
    _ka_check(img, ka.expected_image([((0, 0, 0), 0.1, 0.1, 0.0, (0.9, 0.2, 0.6, 0.5)), ((0, 0, 1), 0.1, 0.1, 0.0, (0.1, 0.8, 0.3, 0.7))]))


@pytest.mark.parametrize("phi", [30.0, -20.0, 75.0])
def test_known_answers_rotated_anisotropic(gsx, pkg, phi):
    rgba = (0.2, 0.6, 1.0, 0.9)
    s = ka.splat((0.2, -0.1, 0.5), ka.cov_xy(0.2, 0.05, 0.1, phi), rgba)[None]
    _ka_check(_ka_render(gsx, pkg, s, [0]), ka.expected_image([((0.2, -0.1, 0.5), 0.2, 0.05, phi, rgba)]))


def test_known_answers_threshold_and_bayer_permutations(gsx, pkg):
    s = ka.splat((0, 0, 0), np.eye(3) * 0.01, (0.3, 0.9, 0.5, 1.0))[None]
    img = _ka_render(gsx, pkg, s, [0], bayerSize=10, bayerScale=0.5)
    want = ka.expected_image([((0, 0, 0), 0.1, 0.1, 0.0, (0.3, 0.9, 0.5, 1.0))], mode="threshold", bayer_scale=0.5)
    assert np.array_equal(img[..., :3], ka.to_rgba8(want))
    # per-instance Bayer permutation (bayerSize 2): `state` clipped splats occupy the first draw ranks.  They sit BEHIND the
    # camera at growing distance so that the sort puts them first (farthest), in index order reversed.
    flat = lambda opacity: ka.splat((0, 0, 0), np.eye(3) * 4.0, (1.0, 1.0, 1.0, opacity))
    for state in sorted(ka.PERMUTATIONS):
        drawn = np.zeros((4, 4), np.int32)
        culled = [ka.splat((0.0, 0.0, 12.0 + k), np.eye(3) * 0.01, (1, 1, 1, 1)) for k in range(state)]   # distance 7 + k > 5
        for opacity in (0.3, 0.5, 0.7, 0.9):
            s = np.stack(culled + [flat(opacity)])
            img = _ka_render(gsx, pkg, s, list(range(state - 1, -1, -1)) + [state], bayerSize=2)
            block = img[48:52, 48:52, 0]
            assert set(np.unique(block)) <= {13, 255}
            drawn += (block == 255)
        value = 4 - drawn
        arr = ka.PERMUTATIONS[state]
        for y in range(4):
            for x in range(4):
                assert value[y, x] == arr[(x % 2) * 2 + (y % 2)], (state, x, y)
    # bayerSize 4, state 0: value = 16 - number of the 16 frames (opacity (k + 0.5) / 17) in which the pixel survives
    drawn = np.zeros((4, 4), np.int32)
    for k in range(1, 17):
        img = _ka_render(gsx, pkg, flat((k + 0.5) / 17.0)[None], [0], bayerSize=4)
        drawn += (img[48:52, 48:52, 0] == 255)
    for x in range(4):
        for y in range(4):
            assert 16 - drawn[y, x] == ka.BAYER4_STATE0[x][y], (x, y)


def test_dense_scene_default_schedule_matches_oracle(gsx_fresh, pkg, oracle, synth):
    """The headline code path — DEFAULT options, history-driven schedule, several iterations, tiles retiring, per-pixel
    freeze — against the image oracle: 420 k opaque-ish splats at 640x360 (first window 256 K ranks, so the frame
    takes at least two iterations; the oracle needs a few seconds)."""
    gsx = gsx_fresh
    w, h, n = 640, 360, 420_000
    splats = oracle.raw_to_splat(synth.clustered_cloud(n, seed=17, n_clusters=6, sigma=0.5, frac=0.8, centre_extent=1.0, scale_mul=1.5))
    view, proj, cam = camera(oracle, w, h, steps=1)
    u = pkg.make_uniforms(proj, view)
    gsx.upload(splats)
    perm = gsx.sort(cam)
    ref = oracle.render(splats, perm, orc.make_uniforms(proj, view), w, h)
    assert ref[..., :3].std() > 5
    frames = []
    for k in range(4):                                   # frame 0 has no history; 1 .. 3 run the history-driven schedule
        frames.append(gsx.render(u, w, h).copy())
        t = gsx.timings()
        if k == 0:
            assert t["render_iters"] >= 2, t
            assert t["tile_pairs"] > 1_000_000
    check_image(frames[0], ref)
    for f in frames[1:]:
        assert np.array_equal(f, frames[0])
    # a second pose on the same context (the schedule now carries the first pose's history)
    view2, proj2, cam2 = camera(oracle, w, h, steps=7)
    perm2 = gsx.sort(cam2)
    img2 = gsx.render(pkg.make_uniforms(proj2, view2), w, h)
    pairs_default = gsx.timings()["tile_pairs"]
    check_image(img2, oracle.render(splats, perm2, orc.make_uniforms(proj2, view2), w, h))
    # tiles did retire: binning everything in one shot (and without per-tile rank caps) emits more pairs, and the same bytes
    gsx.set_option("tile_caps", 0)
    gsx.set_option("chunk_first", 1 << 30)
    one = gsx.render(pkg.make_uniforms(proj2, view2), w, h)
    assert gsx.timings()["tile_pairs"] > pairs_default
    assert np.array_equal(one, img2)


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


def test_scatter_slot_options_change_nothing(gsx_fresh, pkg, oracle, synth):
    """gsx_set_option("sort_slots" | "pair_slots"): fewer resident scatter workgroups (longer column ranges in both radix
    sorts) give the same permutation and the same frame, byte for byte; out-of-range values are rejected."""
    gsx = gsx_fresh
    w, h = 640, 360
    splats = oracle.raw_to_splat(synth.clustered_cloud(300_000, seed=21))
    view, proj, cam = camera(oracle, w, h)
    u = pkg.make_uniforms(proj, view)
    gsx.upload(splats)
    want_perm = gsx.sort(cam).copy()
    want = gsx.render(u, w, h).copy()
    for ss, ps in ((8, 8), (24, 40), (2048, 2048)):
        gsx.set_option("sort_slots", ss)
        gsx.set_option("pair_slots", ps)
        assert np.array_equal(gsx.sort(cam), want_perm), (ss, ps)
        assert np.array_equal(gsx.render(u, w, h), want), (ss, ps)
    for bad in (0, 7, 2049):
        with pytest.raises(pkg.GsxError):
            gsx.set_option("sort_slots", bad)


def test_large_render_properties(gsx_fresh, pkg, oracle, synth):
    """cfg-2 size (6M splats, 1920x1080): the image oracle would take minutes for the whole frame, so check size-independent
    properties of the frame loop, and the oracle on one 64-row band of it."""
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
    # (4b) the image oracle at FULL size on a band of the frame: rows 508 .. 572 of the 1080p picture (four tile rows) with all 6 M splats
    #      (the oracle projects every splat and rasterizes the band's rows: ~10 s of CPU)
    yb0, yb1 = 508, 572
    ref_band = oracle.render(splats, perm, orc.make_uniforms(proj, view), w, h, yb0, yb1)
    assert ref_band[yb0:yb1, :, :3].std() > 5
    check_image(img[yb0:yb1], ref_band[yb0:yb1])
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


def test_cfg4_all_64_poses_at_20m(gsx_fresh, pkg, oracle, synth):
    """cfg 4 at its real shape: the 20 M-splat scene under all 64 poses of the pose set (8 azimuths x 8 elevations, what
    bench.py --shard views deals out).  Every pose: the permutation is sorted by the oracle's keys, ties ascend by index,
    every index appears once; every 8th pose: a full permutation check and frame determinism (two renders, same bytes);
    one pose: the frame against the image oracle at 96 x 54 (the oracle projects all 20 M splats: ~25 s of CPU)."""
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    gsx = gsx_fresh
    n, w, h = 20_000_000, 1920, 1080
    splats = oracle.raw_to_splat(synth.clustered_cloud(n, seed=20))
    gsx.upload(splats)
    views = bench.orbit_views(pkg, 64, w, h)
    total = n * (n - 1) // 2
    forms = set()
    for v, (view, proj, cam) in enumerate(views):
        perm = gsx.sort(cam)
        forms.add(gsx.timings()["sort_form"])
        assert len(perm) == n
        k = oracle.distance_keys(splats, cam)[perm]
        assert np.all(k[:-1] >= k[1:]), v                                 # sortedness
        same = k[1:] == k[:-1]
        assert np.all(perm[1:][same] > perm[:-1][same]), v                # ties by ascending index
        assert int(perm.sum(dtype=np.uint64)) == total, v                 # (cheap necessary condition of a permutation)
        if v % 8 == 0:
            assert np.array_equal(np.bincount(perm, minlength=n), np.ones(n, np.int64)), v
            u = pkg.make_uniforms(proj, view)
            img = gsx.render(u, w, h)
            assert np.all(img[..., 3] == 255) and img[..., :3].std() > 10
            assert np.array_equal(gsx.render(u, w, h), img), v
    assert forms <= {1, 2}
    view, proj, cam = views[5]
    perm = gsx.sort(cam)
    ws, hs = 96, 54
    img = gsx.render(pkg.make_uniforms(proj, view), ws, hs)
    check_image(img, oracle.render(splats, perm, orc.make_uniforms(proj, view), ws, hs))
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
    gsx.set_option("sample_every", 4)
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
    # six sorts since set_profiling(2): every 4th is sampled (the 1st and the 5th): the first brackets every pass of the sort,
    # the second keys + sort as a whole
    passes = gsx.timings()["sort_passes"]
    assert set(launches) == set(range(passes)) and all(v == 1 for v in launches.values()) and whole_n == 1
    assert all(v > 0.0 for v in means.values()) and whole_ms > max(means.values())   # the whole takes longer than one of its launches
    for cam3 in [c[2] for c in cams] * 3:                    # nine more sorts: samples at the 9th (passes) and the 13th (whole)
        gsx.sort(cam3, download=False)
    m2, l2 = gsx.scatter_samples()
    whole_ms, whole_n = gsx.sort_samples()
    assert set(l2) == set(range(passes)) and all(v == 1 for v in l2.values()) and whole_n == 1
    assert whole_ms > max(m2.values())


def test_sort_ahead_draws_each_pose_from_its_own_permutation(gsx_fresh, pkg, oracle, synth):
    # option sort_ahead: the caller queues frame k+1's sort BEFORE frame k; gsx_render draws the oldest permutation not drawn
    # yet, so every frame is byte-identical to the in-order call sequence (sort k, render k)
    gsx = gsx_fresh
    w, h = 240, 176
    splats = dense_scene(oracle, synth, 20000, seed=7)
    gsx.upload(splats)
    cams = [camera(oracle, w, h, steps=s) for s in (0, 3, 6, 9, 12)]
    unis = [pkg.make_uniforms(proj, view) for view, proj, _ in cams]
    ref = []
    for (view, proj, cam), u in zip(cams, unis):
        gsx.sort(cam, download=False)
        ref.append(gsx.render(u, w, h).copy())
    assert not np.array_equal(ref[0], ref[1]) and not np.array_equal(ref[1], ref[2])
    gsx.set_option("sort_ahead", 1)
    gsx.sort(cams[0][2], download=False)
    for k in range(len(cams)):
        if k + 1 < len(cams):
            gsx.sort(cams[k + 1][2], download=False)       # the NEXT pose's sort first ...
        gsx.render(unis[k], w, h, download=False)          # ... then this pose's frame, from the permutation sorted one step ago
        if k % 2:                                          # (every other frame is left in flight while the next two calls are queued)
            gsx.sync()
        ptr, fw, fh = gsx.framebuffer_dev()
        if k % 2:
            assert np.array_equal(_read_device(ptr, w * h * 4).reshape(h, w, 4), ref[k]), f"sort_ahead: frame {k} differs"
    gsx.sync()
    ptr, _, _ = gsx.framebuffer_dev()
    assert np.array_equal(_read_device(ptr, w * h * 4).reshape(h, w, 4), ref[-1])
    # a third sort before any frame: only two permutations wait, the oldest of the three is dropped
    for k in (0, 1, 2):
        gsx.sort(cams[k][2], download=False)
    assert np.array_equal(gsx.render(unis[1], w, h), ref[1])
    assert np.array_equal(gsx.render(unis[2], w, h), ref[2])
    assert np.array_equal(gsx.render(unis[2], w, h), ref[2])      # nothing waits any more: the latest one again
    # setting the option forgets what waits; without it the latest sort wins as before
    gsx.sort(cams[3][2], download=False)
    gsx.set_option("sort_ahead", 0)
    gsx.sort(cams[4][2], download=False)
    gsx.sort(cams[0][2], download=False)
    assert np.array_equal(gsx.render(unis[0], w, h), ref[0])


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
    gsx.set_framebuffer_dev(None)                             # (queued frames keep the target they were started on)
    gsx.sync()
    host = buf.cpu().numpy()
    for slot in (0, 1):
        assert np.array_equal(host[slot * nbytes:(slot + 1) * nbytes].reshape(h, w, 4), img)
    assert np.all(host[2 * nbytes:] == 7)
    # the library's stream is a usable torch stream: work queued on it is ordered behind the frame
    gsx.set_framebuffer_dev(buf.data_ptr(), nbytes)
    gsx.render(u, w, h, download=False)
    gsx.set_framebuffer_dev(None)
    # (gsx_get_stream orders the main stream behind every frame queued so far: frames alternate between two streams)
    ext = torch.cuda.ExternalStream(gsx.stream(), device=torch.device("cuda", 0))
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


@pytest.mark.parametrize("lod", [False, True])
def test_two_stream_pipeline_matches_single_stream(gsx_fresh, pkg, oracle, synth, lod):
    """The LOD cut and the depth sort of frame k+1 run on a second stream beside the binning / rasterizing of frame k
    (option async_sort, default on), and two frames are in flight at a time on two more (option frames_in_flight, default 2).
    Twelve frames of an orbit queued without a single read-back, each into its own slot of a
    caller-owned buffer, must be byte-identical to the same twelve frames with everything on one stream - also with
    profiling level 2 on, which moves every 8th sort to the main stream."""
    import importlib
    lodm = importlib.import_module("gaussian_splatting_amd.lod")
    gsx = gsx_fresh
    w, h, frames = 480, 270, 12
    nbytes = w * h * 4
    tree = None
    if lod:
        tree = lodm.build_octree(synth.clustered_cloud(300_000, seed=9))
        tree.upload(gsx)
    else:
        gsx.upload(oracle.raw_to_splat(synth.clustered_cloud(300_000, seed=9)))
    poses = []
    for f in range(frames):
        world, _ = oracle.orbit_world(4.0, (0.0, 1.0, 0.0), 2.0 * np.pi / 40.0, f)
        view, proj, vp, cam = oracle.camera(world, 45.0, w / h)
        poses.append((pkg.make_uniforms(proj, view), cam, vp))
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    buf = C.c_void_p()
    assert hip.hipMalloc(C.byref(buf), frames * nbytes) == 0

    def run(async_on, level):
        gsx.set_option("async_sort", async_on)
        gsx.set_option("frames_in_flight", 2 if async_on else 1)
        gsx.set_profiling(level)
        gsx.sync()
        assert hip.hipMemset(buf, 0, frames * nbytes) == 0
        for f, (u, cam, vp) in enumerate(poses):
            if lod:
                gsx.select_octree(vp, 0.002, want_count=False)
            gsx.sort(cam, download=False)
            gsx.set_framebuffer_dev(buf.value + f * nbytes, nbytes)
            gsx.render(u, w, h, download=False)
        gsx.set_framebuffer_dev(None)
        gsx.sync()
        return _read_device(buf.value, frames * nbytes).reshape(frames, h, w, 4).copy()

    ref = run(0, 0)
    assert ref[..., :3].std() > 5 and not np.array_equal(ref[0], ref[6])
    for level in (0, 2):
        got = run(1, level)
        for f in range(frames):
            assert np.array_equal(got[f], ref[f]), (level, f)
    gsx.set_profiling(0)
    assert hip.hipFree(buf) == 0
    if tree is not None:
        gsx.select_all()
        tree.close()


def test_tile_rank_caps_same_pixels_fewer_pairs_and_redo(gsx_fresh, pkg, oracle, synth):
    """Per-tile rank caps (option tile_caps, default on): with a steady view every tile is binned only up to 1.5 x the depth at
    which it saturated in the frame slot's last frame.  Same bytes as without caps, far fewer pairs; and when the prediction
    is wrong - here the scene grows eightfold between two frames of the same view, so tiles have to look much deeper than
    their caps - the withheld pairs are supplied by a redo continuation and the bytes are still the same."""
    gsx = gsx_fresh
    w, h, n = 960, 540, 1_500_000
    splats = oracle.raw_to_splat(synth.clustered_cloud(n, seed=6))
    view, proj, cam = camera(oracle, w, h)
    u = pkg.make_uniforms(proj, view)
    gsx.upload(splats)
    perm = gsx.sort(cam)
    gsx.set_option("tile_caps", 0)
    ref = gsx.render(u, w, h).copy()
    pairs_plain = gsx.timings()["tile_pairs"]
    assert gsx.timings()["render_capped"] == 0
    gsx.set_option("tile_caps", 1)
    capped = 0
    for k in range(6):                                       # both frame slots need a finished frame of this view first
        gsx.sort(cam, download=False)
        img = gsx.render(u, w, h)
        t = gsx.timings()
        assert np.array_equal(img, ref), k
        if t["render_capped"]:
            capped += 1
            assert t["render_redos"] == 0
            assert t["tile_pairs"] < 0.9 * pairs_plain, (t["tile_pairs"], pairs_plain)
    assert capped >= 3
    # a prediction that is wrong: tile_caps = 2 caps every tile at 1/8 of the depth it needed last frame, so nearly every tile
    # that saturated before is starved now and gets the withheld pairs from a redo continuation
    gsx.set_option("tile_caps", 2)
    gsx.sort(cam, download=False)
    img = gsx.render(u, w, h)
    t = gsx.timings()
    assert t["render_capped"] == 1 and t["render_redos"] == 1, (t["render_capped"], t["render_redos"])
    assert np.array_equal(img, ref)
    for k in range(4):                                       # a failed prediction pauses the caps (4 frames the first time)
        gsx.sort(cam, download=False)
        img = gsx.render(u, w, h)
        assert gsx.timings()["render_capped"] == 0 and np.array_equal(img, ref), k
    gsx.sort(cam, download=False)
    img = gsx.render(u, w, h)                                # ... then they are tried again (and fail again: still 1/8)
    t = gsx.timings()
    assert t["render_capped"] == 1 and t["render_redos"] == 1 and np.array_equal(img, ref)
    gsx.set_option("tile_caps", 1)
    # ... and one that is wrong because the scene changed under a steady view: frames of every 8th splat, then all of them
    gsx.select_list(np.arange(0, n, 8, dtype=np.uint32))
    for k in range(12):                                      # (long enough for the pause above to run out)
        gsx.sort(cam, download=False)
        sparse = gsx.render(u, w, h)
    gsx.select_all()
    gsx.sort(cam, download=False)
    assert np.array_equal(gsx.render(u, w, h), ref)
    assert not np.array_equal(sparse, ref)
    # a view that moved by more than a few tiles switches the caps off by itself
    view2, proj2, cam2 = camera(oracle, w, h, steps=3)
    gsx.sort(cam2, download=False)
    gsx.render(pkg.make_uniforms(proj2, view2), w, h)
    assert gsx.timings()["render_capped"] == 0


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_call_sequences_match_the_strict_order(pkg, oracle, synth, seed):
    """Two contexts get the same random sequence of calls - selections (all / list / octree cut), sorts and frames with and
    without read-back, tile bands, render targets, option changes, syncs.  One runs with everything the library overlaps or
    predicts switched on (sort stream, two frames in flight, rank caps, history), the other strictly in order with all of it
    off.  Every permutation and every frame read back must agree byte for byte: orderings between the streams that the
    dedicated tests do not think of show up here."""
    import importlib
    lodm = importlib.import_module("gaussian_splatting_amd.lod")
    rng = np.random.default_rng(seed)
    w, h = 320, 208
    tree = lodm.build_octree(synth.clustered_cloud(120_000, seed=seed))
    n = tree.n
    fast, strict = pkg.Gsx(0), pkg.Gsx(0)
    try:
        for g in (fast, strict):
            tree.upload(g)
        for k, v in (("async_sort", 0), ("frames_in_flight", 1), ("tile_caps", 0), ("no_history", 1)):
            strict.set_option(k, v)
        if seed == 4:                        # the two forms of the depth sort against each other (cuts and lists of any size)
            fast.set_option("sort_algo", 2)
            strict.set_option("sort_algo", 1)
        poses = []
        for f in range(6):
            world, _ = oracle.orbit_world(4.0, (0.0, 1.0, 0.0), 0.004, f)          # a slow orbit: the rank caps stay in play
            view, proj, vp, cam = oracle.camera(world, 45.0, w / h)
            poses.append((pkg.make_uniforms(proj, view), cam, vp))
        sorted_once = False
        frames = perms = 0
        for step in range(400):
            op = rng.choice(["sort"] * 6 + ["render"] * 9 + ["band"] * 2 + ["sel_all", "sel_list", "sel_oct", "sync", "prof", "timings"])
            u, cam, vp = poses[int(rng.integers(0, len(poses)))] if rng.random() < 0.3 else poses[step // 100]
            if op == "sort":
                if rng.random() < 0.3:
                    a, b = fast.sort(cam), strict.sort(cam)
                    assert np.array_equal(a, b), step
                    perms += 1
                else:
                    fast.sort(cam, download=False); strict.sort(cam, download=False)
                sorted_once = True
            elif op == "render" and sorted_once:
                if rng.random() < 0.4:
                    a, b = fast.render(u, w, h), strict.render(u, w, h)
                    assert np.array_equal(a, b), step
                    frames += 1
                else:
                    fast.render(u, w, h, download=False); strict.render(u, w, h, download=False)
            elif op == "band" and sorted_once:
                y0 = int(rng.integers(0, h - 16)); y1 = int(rng.integers(y0 + 1, h + 1))
                a, b = fast.render(u, w, h, y0, y1), strict.render(u, w, h, y0, y1)
                assert np.array_equal(a[y0:y1], b[y0:y1]), step
            elif op == "sel_all":
                fast.select_all(); strict.select_all(); sorted_once = False
            elif op == "sel_list":
                idx = rng.choice(n, size=int(rng.integers(1, n)), replace=False).astype(np.uint32)
                fast.select_list(idx); strict.select_list(idx); sorted_once = False
            elif op == "sel_oct":
                thr = float(rng.choice([0.002, 0.0005, 0.02]))
                want = rng.random() < 0.5
                a, b = fast.select_octree(vp, thr, want_count=want), strict.select_octree(vp, thr, want_count=want)
                if want:
                    assert a == b, step
                sorted_once = False
            elif op == "sync":
                fast.sync(); strict.sync()
            elif op == "prof":
                fast.set_profiling(int(rng.choice([0, 0, 2, 1])))
            elif op == "timings" and sorted_once:
                fast.timings(); strict.timings()
        # a final frame of every pose, read back
        for u, cam, vp in poses:
            assert np.array_equal(fast.sort(cam), strict.sort(cam))
            assert np.array_equal(fast.render(u, w, h), strict.render(u, w, h))
        assert frames >= 10 and perms >= 5, (frames, perms)
    finally:
        fast.close(); strict.close(); tree.close()


@pytest.mark.parametrize("seed", [11, 12])
def test_random_call_sequences_with_sort_ahead(pkg, oracle, synth, seed):
    """Option sort_ahead under random call sequences: `fast` queues sorts ahead of frames (up to two permutations wait, a third
    sort drops the oldest, gsx_render draws the oldest), `strict` runs in order with everything else switched off and is told
    which camera each frame belongs to.  Every frame read back must agree byte for byte."""
    import importlib
    lodm = importlib.import_module("gaussian_splatting_amd.lod")
    rng = np.random.default_rng(seed)
    w, h = 320, 208
    tree = lodm.build_octree(synth.clustered_cloud(150_000, seed=seed))
    n = tree.n
    fast, strict = pkg.Gsx(0), pkg.Gsx(0)
    try:
        for g in (fast, strict):
            tree.upload(g)
        for k, v in (("async_sort", 0), ("frames_in_flight", 1), ("tile_caps", 0), ("no_history", 1), ("sort_algo", 1)):
            strict.set_option(k, v)
        fast.set_option("sort_algo", 2 if seed % 2 else 0)
        fast.set_option("sort_ahead", 1)
        poses = []
        for f in range(8):
            world, _ = oracle.orbit_world(4.0, (0.0, 1.0, 0.0), 0.02, f)
            view, proj, vp, cam = oracle.camera(world, 45.0, w / h)
            poses.append((pkg.make_uniforms(proj, view), cam, vp))
        waiting, latest, frames = [], None, 0            # poses whose permutation waits to be drawn in `fast`; the latest sort
        for step in range(300):
            op = rng.choice(["sort"] * 8 + ["render"] * 8 + ["sel_all", "sel_list", "sel_oct", "sync", "prof"])
            k = int(rng.integers(0, len(poses)))
            if op == "sort":
                fast.sort(poses[k][1], download=False)
                latest = k
                waiting = (waiting + [k])[-2:]
            elif op == "render" and latest is not None:
                draw = waiting.pop(0) if waiting else latest
                u, cam, _ = poses[draw]
                strict.sort(cam, download=False)
                if rng.random() < 0.5:
                    a, b = fast.render(u, w, h), strict.render(u, w, h)
                    assert np.array_equal(a, b), (step, draw)
                    frames += 1
                else:
                    fast.render(u, w, h, download=False); strict.render(u, w, h, download=False)
            elif op in ("sel_all", "sel_list", "sel_oct"):
                if op == "sel_all":
                    fast.select_all(); strict.select_all()
                elif op == "sel_list":
                    idx = rng.choice(n, size=int(rng.integers(1, n)), replace=False).astype(np.uint32)
                    fast.select_list(idx); strict.select_list(idx)
                else:
                    thr = float(rng.choice([0.002, 0.0005, 0.02]))
                    fast.select_octree(poses[k][2], thr, want_count=False); strict.select_octree(poses[k][2], thr, want_count=False)
                fast.set_option("sort_ahead", 1)            # a new selection: forget the permutations of the old one
                waiting, latest = [], None
            elif op == "sync":
                fast.sync(); strict.sync()
            elif op == "prof":
                fast.set_profiling(int(rng.choice([0, 0, 2])))
        assert frames >= 20, frames
    finally:
        fast.close(); strict.close(); tree.close()


def test_render_into_pinned_frame_matches(gsx, pkg, synth, oracle):
    """Gsx.render(out=pinned_frame): the same bytes as into an ordinary array (the read-back goes by DMA into page-locked memory)."""
    cam = __import__("importlib").import_module("gaussian_splatting_amd.camera")
    splats = oracle.raw_to_splat(synth.clustered_cloud(30000, seed=41), center=True)
    gsx.upload(splats)
    w, h = 320, 200
    pin = pkg.pinned_frame(h, w)
    for az in (0.3, 1.7):
        world = cam.orbit_world(4.0, az, 0.2)
        view, proj, pos = cam.camera_matrices(world, 45.0, w / h)
        u = pkg.make_uniforms(proj, view)
        gsx.sort(pos, download=False)
        ref = gsx.render(u, w, h)
        gsx.sort(pos, download=False)
        got = gsx.render(u, w, h, out=pin)
        assert got is pin and np.array_equal(ref, pin) and pin[..., :3].std() > 3


def test_deferred_readback_delivers_each_frame_two_calls_late(pkg, synth, oracle):
    """Option deferred_readback: gsx_render(host buffer) returns at once and the frame of call k is in ITS buffer when call k + 2
    returns, or after gsx_sync; the bytes are those of the synchronous call."""
    cam = __import__("importlib").import_module("gaussian_splatting_amd.camera")
    splats = oracle.raw_to_splat(synth.clustered_cloud(60000, seed=43), center=True)
    w, h = 400, 240
    g = pkg.Gsx(0)
    try:
        g.upload(splats)
        poses = []
        for k in range(7):
            world = cam.orbit_world(4.0, 0.35 * k, 0.1 * k)
            view, proj, pos = cam.camera_matrices(world, 45.0, w / h)
            poses.append((pos, pkg.make_uniforms(proj, view, bayerSize=(2 if k == 4 else 1))))
        refs = []
        for pos, u in poses:
            g.sort(pos, download=False)
            refs.append(g.render(u, w, h).copy())
        assert len({r.tobytes() for r in refs}) == len(refs)
        g.set_option("deferred_readback", 1)
        ring = [pkg.pinned_frame(h, w) for _ in range(3)]
        for b in ring:
            b[...] = 77
        for k, (pos, u) in enumerate(poses):
            g.sort(pos, download=False)
            g.render(u, w, h, out=ring[k % 3])
            if k >= 2:
                assert np.array_equal(ring[(k - 2) % 3], refs[k - 2]), k
        g.sync()
        assert np.array_equal(ring[5 % 3], refs[5]) and np.array_equal(ring[6 % 3], refs[6])
        # a band: only its rows are written
        ring[0][...] = 77
        g.sort(poses[1][0], download=False)
        g.render(poses[1][1], w, h, 48, 112, out=ring[0])
        g.set_option("deferred_readback", 0)                       # delivers what is queued
        assert np.array_equal(ring[0][48:112], refs[1][48:112]) and np.all(ring[0][:48] == 77) and np.all(ring[0][112:] == 77)
        g.sort(poses[2][0], download=False)
        assert np.array_equal(g.render(poses[2][1], w, h), refs[2])   # synchronous again
    finally:
        g.close()
