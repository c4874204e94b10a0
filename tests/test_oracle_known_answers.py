"""CPU: oracle/orc_raster.cpp against HAND-DERIVED answers (tests/known_answers.py) — the image oracle has no
reference fixture (the reference's WebGPU backend cannot run here), so its every stage is pinned on vectors worked out
from the WGSL program's math: vertex positions, quad extent and the cutOff edge, the falloff, the blend state and the
draw order, a rotated anisotropic splat, the threshold mode and the per-instance Bayer permutation."""
import numpy as np
import pytest

import known_answers as ka
import orc


def uniforms(**kw):
    view, proj = ka.matrices()
    return orc.make_uniforms(proj, view, **kw)


def test_vs_main_quad_vertices(oracle):
    # isotropic sigma = 0.1 at the origin: covariance 0.01 I -> l1 = l2 = 0.01, phi = atan2(0, 0) / 2 = 0, so the quad is
    # centre + 3.5 * 0.1 * (+-1, +-1) = (+-0.35, +-0.35, -5) in view space; clip = (2x, 2y, P10 z + P14, -z)
    s = ka.splat((0, 0, 0), np.eye(3) * 0.01, (1, 1, 1, 1))
    clip = oracle.vs_main(s, uniforms())
    cz = ka.FAR / (ka.NEAR - ka.FAR) * -5.0 - ka.FAR * ka.NEAR / (ka.FAR - ka.NEAR)        # 4.9049049...
    want = np.array([[-0.7, -0.7, cz, 5.0], [0.7, -0.7, cz, 5.0], [0.7, 0.7, cz, 5.0], [-0.7, 0.7, cz, 5.0]])   # SplatMesh.h:171-176 order
    assert np.allclose(clip, want, rtol=0, atol=2e-6)
    # off-centre and nearer: centre (1, -0.5, 1) -> view (1, -0.5, -4), clip.w = 4; sigma 0.2 -> half extent 0.7
    s = ka.splat((1.0, -0.5, 1.0), np.eye(3) * 0.04, (1, 1, 1, 1))
    clip = oracle.vs_main(s, uniforms())
    assert np.allclose(clip[:, 3], 4.0) and np.allclose(clip[0, :2], [2 * (1 - 0.7), 2 * (-0.5 - 0.7)], atol=2e-6)
    assert np.allclose(clip[2, :2], [2 * (1 + 0.7), 2 * (-0.5 + 0.7)], atol=2e-6)
    # splatSize and cutOff scale the offset (wgsl:63): 0.1 * 2.0 * 0.5 = 0.1
    clip = oracle.vs_main(ka.splat((0, 0, 0), np.eye(3) * 0.01, (1, 1, 1, 1)), uniforms(cutOff=2.0, splatSize=0.5))
    assert np.allclose(clip[2, :2], [0.2, 0.2], atol=2e-6)


def test_isotropic_splat_centre_alpha_extent_and_cutoff_edge(oracle):
    # sigma 0.1 at the origin, colour (0.9, 0.4, 0.1), opacity 0.8.  Screen centre (50, 50); quad half extent
    # 3.5 * 0.1 * 100 / 5 = 7 px: pixel centres 43.5 .. 56.5 are inside, i.e. pixels 43 .. 56 in x and y (196 pixels);
    # u = (px + 0.5 - 50) / 7 * 3.5 = (px - 49.5) / 2.
    s = ka.splat((0, 0, 0), np.eye(3) * 0.01, (0.9, 0.4, 0.1, 0.8))[None]
    img, fb = oracle.render(s, np.array([0], np.uint32), uniforms(), ka.W, ka.H, want_f32=True)
    touched = np.any(img[..., :3] != 13, axis=-1) | (np.abs(fb[..., 0] - 0.05) > 1e-7)
    ys, xs = np.nonzero(touched)
    assert (xs.min(), xs.max(), ys.min(), ys.max()) == (43, 56, 43, 56)
    # pixel (50, 50): u = v = 0.25 -> alpha = 0.8 exp(-0.0625) = 0.75153045; rgb = c alpha + 0.05 (1 - alpha)
    assert np.allclose(fb[50, 50, :3], [0.68880088, 0.31303566, 0.08757652], atol=2e-6)
    assert tuple(img[50, 50]) == (176, 80, 22, 255)
    # the cutOff edge: pixel (43, 50) has u = -3.25 (inside, alpha = 0.8 exp(-(3.25^2 + 0.25^2)/2) = 0.00394367)
    assert np.allclose(fb[50, 43, :3], [0.05335212, 0.05138028, 0.05019718], atol=2e-6)
    assert tuple(img[50, 43]) == (14, 13, 13, 255)
    # ... and pixel (42, 50) would have u = -3.75 > cutOff: outside the quad, clear colour 0.05 -> round(12.75) = 13
    assert fb[50, 42, 0] == np.float32(0.05) and tuple(img[50, 42]) == (13, 13, 13, 255)
    # the corner pixel (56, 43): u = 3.25, v = -3.25, alpha = 0.8 exp(-10.5625) = 2.07e-5
    assert np.allclose(fb[43, 56, :3], [0.05001759, 0.05000724, 0.05000103], atol=2e-6)
    # an off-axis pixel (49, 52): u = -0.25, v = 1.25 -> alpha = 0.8 exp(-0.8125) = 0.35499785
    assert np.allclose(fb[52, 49, :3], [0.35174817, 0.17424925, 0.06774989], atol=2e-6)
    assert tuple(img[52, 49]) == (90, 44, 17, 255)
    # the whole frame against the closed form
    want = ka.expected_image([((0, 0, 0), 0.1, 0.1, 0.0, (0.9, 0.4, 0.1, 0.8))])
    assert np.abs(fb[..., :3] - want).max() < 3e-6
    ok = ~ka.near_rounding_boundary(want)
    assert np.array_equal(img[..., :3][ok], ka.to_rgba8(want)[ok]) and np.all(img[..., 3] == 255)


def test_two_splat_blend_follows_draw_order(oracle):
    # far: origin (clip.w 5, half extent 7 px), colour (0.1, 0.8, 0.3), opacity 0.7
    # near: (0, 0, 1) (clip.w 4, half extent 3.5 * 0.1 * 100 / 4 = 8.75 px), colour (0.9, 0.2, 0.6), opacity 0.5
    # pixel (50, 50): far u = v = 0.25 -> a_f = 0.7 exp(-0.0625) = 0.65758914; near u = v = 0.5 / 8.75 * 3.5 = 0.2 ->
    # a_n = 0.5 exp(-0.04) = 0.48039472
    far = ka.splat((0, 0, 0), np.eye(3) * 0.01, (0.1, 0.8, 0.3, 0.7))
    near = ka.splat((0, 0, 1), np.eye(3) * 0.01, (0.9, 0.2, 0.6, 0.5))
    s = np.stack([far, near])
    # draw order = order[] as given (SplatMesh.h:58: instance i draws splats[sortedIndex[i]]): far first, near on top
    img, fb = oracle.render(s, np.array([0, 1], np.uint32), uniforms(), ka.W, ka.H, want_f32=True)
    #   c_n a_n + (c_f a_f + 0.05 (1 - a_f)) (1 - a_n)
    assert np.allclose(fb[50, 50, :3], [0.47541985, 0.37832430, 0.39963879], atol=2e-6)
    assert tuple(img[50, 50]) == (121, 96, 102, 255)
    # the other order composites the other way round: c_f a_f + (c_n a_n + 0.05 (1 - a_n)) (1 - a_f)
    img2, fb2 = oracle.render(s, np.array([1, 0], np.uint32), uniforms(), ka.W, ka.H, want_f32=True)
    assert np.allclose(fb2[50, 50, :3], [0.22269797, 0.56786571, 0.30486809], atol=2e-6)
    assert tuple(img2[50, 50]) == (57, 145, 78, 255)
    # the near splat is larger on screen: pixel (41, 50) is inside its quad only (41.5 > 50 - 8.75, < 50 - 7)
    assert fb[50, 41, 0] > 0.05 + 1e-4 and np.isclose(fb[50, 41, 1], 0.05 + (0.2 - 0.05) * 0.5 * np.exp(-0.5 * (3.4 ** 2 + 0.2 ** 2)), atol=1e-6)
    for order, draws in (((0, 1), "fn"), ((1, 0), "nf")):
        d = {"f": ((0, 0, 0), 0.1, 0.1, 0.0, (0.1, 0.8, 0.3, 0.7)), "n": ((0, 0, 1), 0.1, 0.1, 0.0, (0.9, 0.2, 0.6, 0.5))}
        want = ka.expected_image([d[k] for k in draws])
        _, f = oracle.render(s, np.array(order, np.uint32), uniforms(), ka.W, ka.H, want_f32=True)
        assert np.abs(f[..., :3] - want).max() < 3e-6
    # the alpha channel keeps the clear value (blend: src * 0 + dst * 1, main.cpp:341-344)
    assert np.all(fb[..., 3] == 1.0)


@pytest.mark.parametrize("phi", [30.0, -20.0, 75.0])
def test_rotated_anisotropic_splat(oracle, phi):
    # principal axes 0.2 along (cos phi, sin phi) and 0.05 across it, 0.1 along the view direction; the view-plane 2x2 block
    # of W S W^T is R diag(0.04, 0.0025) R^T, whose eigen-decomposition (wgsl:28-40) must give the axes back:
    # l1 = 0.04, l2 = 0.0025, angle = atan2(2 a01, a11 - a22) / 2 = phi.
    rgba = (0.2, 0.6, 1.0, 0.9)
    s = ka.splat((0.2, -0.1, 0.5), ka.cov_xy(0.2, 0.05, 0.1, phi), rgba)[None]
    img, fb = oracle.render(s, np.array([0], np.uint32), uniforms(), ka.W, ka.H, want_f32=True)
    want = ka.expected_image([((0.2, -0.1, 0.5), 0.2, 0.05, phi, rgba)])
    assert np.abs(fb[..., :3] - want).max() < 1e-5
    ok = ~ka.near_rounding_boundary(want)
    assert np.array_equal(img[..., :3][ok], ka.to_rgba8(want)[ok])
    # the quad is a rotated rectangle, not its bounding box: pixels inside the box but outside the quad stay clear
    touched = np.abs(fb[..., 0] - 0.05) > 1e-9
    ys, xs = np.nonzero(touched)
    box = (xs.max() - xs.min() + 1) * (ys.max() - ys.min() + 1)
    # quad area in pixels: (2 * 3.5 * 0.2) (2 * 3.5 * 0.05) * (100 / 4.5)^2 = 242 px; alpha underflows nowhere inside it
    assert abs(int(touched.sum()) - 242) <= 12 and box > 1.5 * touched.sum()


def test_threshold_mode(oracle):
    # bayerSize 10: alpha' = alpha > bayerScale ? 1 : 0 (wgsl:131-134).  Opacity 1, bayerScale 0.5: full colour where
    # exp(-r^2/2) > 0.5 <=> r^2 < 2 ln 2, a disc of radius sqrt(2 ln 2) = 1.17741 in uv units = 2.3548 px; untouched elsewhere
    s = ka.splat((0, 0, 0), np.eye(3) * 0.01, (0.3, 0.9, 0.5, 1.0))[None]
    img, fb = oracle.render(s, np.array([0], np.uint32), uniforms(bayerSize=10, bayerScale=0.5), ka.W, ka.H, want_f32=True)
    ys, xs = np.mgrid[0:ka.H, 0:ka.W]
    r2 = ((xs - 49.5) / 2.0) ** 2 + ((ys - 49.5) / 2.0) ** 2
    disc = r2 < 2.0 * np.log(2.0)
    assert disc.sum() == 16                           # pixel centres within 2.3548 px of (50, 50): |offset| in {0.5, 1.5} px on both axes (0.5625 + 0.5625 < 1.386 < 1.5625): the central 4 x 4 block
    assert np.allclose(fb[disc][:, :3], [0.3, 0.9, 0.5], atol=1e-7) and np.all(fb[~disc][:, :3] == np.float32(0.05))
    want = ka.expected_image([((0, 0, 0), 0.1, 0.1, 0.0, (0.3, 0.9, 0.5, 1.0))], mode="threshold", bayer_scale=0.5)
    assert np.array_equal(img[..., :3], ka.to_rgba8(want))


def _flat_splat(opacity):
    # sigma 2: within 2 px of the centre r^2 = 2 (2.5 * 0.025)^2 < 0.008, alpha within 0.4 % of the opacity
    return ka.splat((0, 0, 0), np.eye(3) * 4.0, (1.0, 1.0, 1.0, opacity))


@pytest.mark.parametrize("state", sorted(ka.PERMUTATIONS))
def test_bayer_permutation_per_instance(oracle, state):
    # bayerSize 2, bayerScale 1: a fragment survives iff alpha * 5 >= value + 1 (wgsl:206-208), value = arr[(x%2)*2 + (y%2)]
    # with arr = generate_permutation(instance index % 24).  Opacity 0.3 keeps value 0 only, 0.5 keeps {0, 1}, 0.7 keeps
    # {0, 1, 2}, 0.9 keeps all: value(x, y) = 4 - (number of the four frames in which the pixel is drawn).  The instance index
    # is the draw rank, culled instances included: `state` clipped splats are drawn first.
    drawn = np.zeros((4, 4), np.int32)
    for opacity in (0.3, 0.5, 0.7, 0.9):
        s = np.stack([ka.culled_splat()] * state + [_flat_splat(opacity)])
        img = oracle.render(s, np.arange(state + 1, dtype=np.uint32), uniforms(bayerSize=2), ka.W, ka.H)
        block = img[48:52, 48:52, 0]
        assert set(np.unique(block)) <= {13, 255}          # discarded -> clear colour, kept -> alpha 1 -> white
        drawn += (block == 255)
    value = 4 - drawn                                      # [y][x]
    arr = ka.PERMUTATIONS[state]
    for y in range(4):
        for x in range(4):
            assert value[y, x] == arr[(x % 2) * 2 + (y % 2)], (state, x, y)


def test_bayer4_values_state0(oracle):
    # bayerSize 4: survive iff alpha * 17 >= value + 1; opacities (k + 0.5) / 17, k = 1 .. 16, keep values <= k - 1, so
    # value = 16 - (number of frames in which the pixel is drawn).  Pixels 48 .. 51 are x % 4 = 0 .. 3.
    drawn = np.zeros((4, 4), np.int32)
    for k in range(1, 17):
        s = _flat_splat((k + 0.5) / 17.0)[None]
        img = oracle.render(s, np.array([0], np.uint32), uniforms(bayerSize=4), ka.W, ka.H)
        drawn += (img[48:52, 48:52, 0] == 255)
    value = 16 - drawn
    for x in range(4):
        for y in range(4):
            assert value[y, x] == ka.BAYER4_STATE0[x][y], (x, y)


def test_clip_volume_and_degenerate_inputs(oracle):
    u = uniforms()
    # behind the camera (clip.w < 0) and beyond the far plane (clip.z > clip.w): nothing drawn
    for z in (6.0, -120.0):
        img = oracle.render(ka.splat((0, 0, z), np.eye(3) * 0.01, (1, 1, 1, 1))[None], np.array([0], np.uint32), u, ka.W, ka.H)
        assert np.all(img[..., :3] == 13)
    # nearer than the near plane (view z = -0.05): clip.z < 0
    img = oracle.render(ka.splat((0, 0, 4.95), np.eye(3) * 1e-6, (1, 1, 1, 1))[None], np.array([0], np.uint32), u, ka.W, ka.H)
    assert np.all(img[..., :3] == 13)
    # empty draw list: the clear colour, alpha 1
    img = oracle.render(np.zeros((1, 20), np.float32), np.zeros(0, np.uint32), u, ka.W, ka.H)
    assert np.all(img[..., :3] == 13) and np.all(img[..., 3] == 255)
    # a tile band [y0, y1) of the first known-answer frame equals those rows of the whole frame
    s = ka.splat((0, 0, 0), np.eye(3) * 0.01, (0.9, 0.4, 0.1, 0.8))[None]
    whole = oracle.render(s, np.array([0], np.uint32), u, ka.W, ka.H)
    band = oracle.render(s, np.array([0], np.uint32), u, ka.W, ka.H, 45, 52)
    assert np.array_equal(band[45:52], whole[45:52])
