"""Hand-derived known-answer scenes for the image path (resources/shader_quads_ordered.wgsl + the pipeline's blend
state, main.cpp:338-344, clear colour renderer.cpp:162-164).

Nothing here calls the oracle or the GPU library: expected pixels come from the closed form of the reference's math,
written out below, so the same vectors pin BOTH oracle/orc_raster.cpp (tests/test_oracle_known_answers.py, CPU) and
the HIP path (tests/test_gpu_render.py, GPU).

Set-up shared by all cases (column-major 4x4 like glm / WGSL):
  model = I, view = translate(0, 0, -5): a world point (x, y, z) has view coordinates (x, y, z - 5);
  proj  = diag(2, 2, .) with the last row (0, 0, -1, 0): clip.w = 5 - z, clip.xy = 2 (x, y)
          (P[10] = far/(near-far), P[14] = -far near/(far-near) with near 0.1, far 100: 0 <= clip.z <= clip.w holds)
  frame  = 100 x 100, so a view-space offset d at clip.w = wc moves the pixel by d * 2/wc * 50 = 100 d / wc pixels,
           screen y pointing down.
A splat with view-plane covariance eigenpairs (l1, phi), (l2, phi + 90 deg) is drawn (wgsl:28-40, 63, 70) as the quad
    centre + u * e0 + v * e1,   e0 = sqrt(l1) (cos phi, sin phi),  e1 = sqrt(l2) (-sin phi, cos phi),  |u|, |v| <= cutOff
(splatSize = 1) and uv is interpolated linearly over it, so a pixel at view-space offset d from the centre has
    u = d . e0 / l1,   v = d . e1 / l2      (e0, e1 orthogonal)
    alpha = opacity * exp(-(u^2 + v^2) / 2)                              (wgsl:123-125)
    dst.rgb = src.rgb * alpha + dst.rgb * (1 - alpha)                    (main.cpp:338-344), dst starts at 0.05.
"""
import numpy as np

W = H = 100
NEAR, FAR = 0.1, 100.0
CLEAR = 0.05


def matrices():
    view = np.eye(4, dtype=np.float32)
    view[3, 2] = -5.0                      # row-major numpy [col][row]: translation column, z component
    proj = np.zeros((4, 4), np.float32)
    proj[0, 0] = 2.0
    proj[1, 1] = 2.0
    proj[2, 2] = FAR / (NEAR - FAR)
    proj[2, 3] = -1.0
    proj[3, 2] = -(FAR * NEAR) / (FAR - NEAR)
    return view.reshape(-1), proj.reshape(-1)


def splat(pos, cov3, rgba):
    """80-byte Splat record (Splat.h:12-22): mat4 with the covariance in the upper 3x3 and the position in column 3, colour."""
    s = np.zeros(20, np.float32)
    m = np.zeros((4, 4), np.float32)
    m[:3, :3] = np.asarray(cov3, np.float32).T      # [col][row]; symmetric anyway
    m[3, :3] = pos
    m[3, 3] = 1.0
    s[:16] = m.reshape(-1)
    s[16:] = rgba
    return s


def cov_xy(s1, s2, s3, phi_deg):
    """R_z(phi) diag(s1^2, s2^2, s3^2) R_z(phi)^T: principal axes s1 along (cos phi, sin phi), s2 across, s3 along z."""
    p = np.deg2rad(phi_deg)
    r = np.array([[np.cos(p), -np.sin(p), 0.0], [np.sin(p), np.cos(p), 0.0], [0.0, 0.0, 1.0]])
    return r @ np.diag([s1 * s1, s2 * s2, s3 * s3]) @ r.T


def expected_image(draws, cut=3.5, mode="blend", bayer_scale=1.0, bayer_value=None):
    """Closed-form frame in float64.  draws: [(pos3, s1, s2, phi_deg, rgba)] in DRAW order (first = drawn first = farthest).
    mode "blend" (bayerSize 1), "threshold" (bayerSize 10: alpha = alpha > bayerScale, wgsl:131-134) or "dither"
    (bayer_value(x, y, rank) supplied by the caller; wgsl:203-210)."""
    fb = np.full((H, W, 3), CLEAR, np.float64)
    ys, xs = np.mgrid[0:H, 0:W]
    for rank, (pos, s1, s2, phi, rgba) in enumerate(draws):
        wc = 5.0 - pos[2]
        if wc <= 0:
            continue                                   # behind the camera: clipped
        cx, cy = (pos[0] / wc + 0.5) * W, (0.5 - pos[1] / wc) * H
        dx, dy = (xs + 0.5 - cx) * wc / 100.0, -(ys + 0.5 - cy) * wc / 100.0      # view-space offset, y up
        p = np.deg2rad(phi)
        u = (dx * np.cos(p) + dy * np.sin(p)) / s1
        v = (-dx * np.sin(p) + dy * np.cos(p)) / s2
        inside = (np.abs(u) <= cut) & (np.abs(v) <= cut)
        alpha = rgba[3] * np.exp(-0.5 * (u * u + v * v))
        if mode == "threshold":
            alpha = (alpha > bayer_scale).astype(np.float64)
        elif mode == "dither":
            size = bayer_value.size
            keep = ~((alpha * bayer_scale) * (size * size + 1) < bayer_value(xs, ys, rank) + 1.0)
            inside &= keep
            alpha = np.ones_like(alpha)
        a = np.where(inside, alpha, 0.0)[..., None]
        fb = np.asarray(rgba[:3], np.float64) * a + fb * (1.0 - a)
    return fb


def to_rgba8(fb):
    return np.floor(np.clip(fb, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)


def near_rounding_boundary(fb, margin=0.02):
    """Pixels/channels whose exact value sits within `margin` of an 8-bit rounding boundary (excluded from byte-exact checks)."""
    f = np.clip(fb, 0.0, 1.0) * 255.0 + 0.5
    return np.abs(f - np.round(f)) < margin


# ---- generate_permutation (wgsl:75-91), worked by hand: fac = (s / 6, (s % 6) / 2, s % 2), then for i = 0, 1, 2 swap
#      arr[i] <-> arr[fac[i] + i] starting from (0, 1, 2, 3) ----
#   s = 0 : fac (0,0,0): nothing moves                                          -> 0 1 2 3
#   s = 1 : fac (0,0,1): i=2 swaps arr[2], arr[3]                               -> 0 1 3 2
#   s = 2 : fac (0,1,0): i=1 swaps arr[1], arr[2]                               -> 0 2 1 3
#   s = 3 : fac (0,1,1): i=1 -> 0 2 1 3, i=2 swaps arr[2], arr[3]               -> 0 2 3 1
#   s = 4 : fac (0,2,0): i=1 swaps arr[1], arr[3]                               -> 0 3 2 1
#   s = 5 : fac (0,2,1): i=1 -> 0 3 2 1, i=2 swaps arr[2], arr[3]               -> 0 3 1 2
#   s = 6 : fac (1,0,0): i=0 swaps arr[0], arr[1]                               -> 1 0 2 3
#   s = 13: fac (2,0,1): i=0 swaps arr[0], arr[2] -> 2 1 0 3, i=2 -> 2 1 3 0    -> 2 1 3 0
#   s = 23: fac (3,2,1): i=0 -> 3 1 2 0, i=1 swaps arr[1], arr[3] -> 3 0 2 1, i=2 swaps arr[2], arr[3] -> 3 0 1 2
PERMUTATIONS = {0: (0, 1, 2, 3), 1: (0, 1, 3, 2), 2: (0, 2, 1, 3), 3: (0, 2, 3, 1), 4: (0, 3, 2, 1), 5: (0, 3, 1, 2),
                6: (1, 0, 2, 3), 13: (2, 1, 3, 0), 23: (3, 0, 1, 2)}
PERMUTATIONS[24] = PERMUTATIONS[0]          # generate_bayer takes the state modulo 24 (wgsl:114)
PERMUTATIONS[29] = PERMUTATIONS[5]

# bayerSize 2 (wgsl:93-111, one round): value(x, y) = arr[(x % 2) * 2 + (y % 2)]
# bayerSize 4, state 0 (two rounds: the 2x2 quadrant of the 4x4 cell is the LOW digit, the position inside it the high one):
#   value(x, y) = arr[(x%4 / 2) * 2 + (y%4 / 2)] + 4 * arr[(x % 2) * 2 + (y % 2)], arr = (0, 1, 2, 3); rows are x, columns y
BAYER4_STATE0 = ((0, 4, 1, 5), (8, 12, 9, 13), (2, 6, 3, 7), (10, 14, 11, 15))


class Bayer2:
    size = 2

    def __call__(self, xs, ys, rank):
        arr = np.asarray(PERMUTATIONS[rank])
        return arr[(xs % 2) * 2 + (ys % 2)].astype(np.float64)


def culled_splat():
    """Behind the camera (view z = +1): clipped, but it still occupies a draw rank (instance index)."""
    return splat((0.0, 0.0, 6.0), np.eye(3) * 0.01, (1.0, 1.0, 1.0, 1.0))
