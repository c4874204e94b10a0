"""Synthetic splat clouds in the reference's on-disk format.

Every scene is emitted as 32-byte `SplatRaw` records (reference Splat.h:5-10:
position f32x3, scale f32x3, rgba u8x4, quaternion u8x4 as (w,x,y,z)), i.e. the
`.splat` file layout ResourceManager::loadSplatsRaw streams
(ResourceManager.cpp:36-65), so data always passes through the reference's own
u8 quantisation.  Scene recipes follow SURVEY.md section 8(d).
"""
import numpy as np


def _pack(pos, scale, rgba, quat):
    n = pos.shape[0]
    raw = np.empty((n, 32), np.uint8)
    raw[:, 0:12] = np.ascontiguousarray(pos, np.float32).view(np.uint8).reshape(n, 12)
    raw[:, 12:24] = np.ascontiguousarray(scale, np.float32).view(np.uint8).reshape(n, 12)
    raw[:, 24:28] = rgba
    raw[:, 28:32] = quat
    return raw


def _attributes(rng, n, scale_mul=1.0):
    lo, hi = np.log(0.005), np.log(0.05)
    scale = np.exp(rng.uniform(lo, hi, size=(n, 3))).astype(np.float32)
    # keep every splat anisotropic (max/min >= 1.2) so the 2x2 eigen-decomposition
    # (wgsl:28-40) stays off its isotropic knife edge
    mn, mx = scale.min(axis=1), scale.max(axis=1)
    fix = (mx / mn) < 1.2
    am = scale.argmax(axis=1)
    scale[np.nonzero(fix)[0], am[fix]] = (mn[fix] * 1.2).astype(np.float32)
    scale *= np.float32(scale_mul)
    q = rng.standard_normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    quat = np.clip(np.rint(q * 128.0 + 128.0), 0, 255).astype(np.uint8)
    rgba = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    rgba[:, 3] = rng.integers(26, 256, size=n, dtype=np.uint8)
    return scale, rgba, quat


def uniform_cloud(n, seed=1, half_extent=3.0):
    """cfg 1: positions uniform in [-3,3]^3."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pos = rng.uniform(-half_extent, half_extent, size=(n, 3)).astype(np.float32)
    return _pack(pos, *_attributes(rng, n))


def clustered_cloud(n, seed=6, n_clusters=64, sigma=0.25, frac=0.7, centre_extent=2.5, scale_mul=1.0):
    """cfg 2/3/4: `frac` of the splats in Gaussian clusters, the rest uniform."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nc = int(n * frac)
    centres = rng.uniform(-centre_extent, centre_extent, size=(n_clusters, 3))
    which = rng.integers(0, n_clusters, size=nc)
    pos = np.empty((n, 3), np.float64)
    pos[:nc] = centres[which] + rng.standard_normal(size=(nc, 3)) * sigma
    pos[nc:] = rng.uniform(-3.0, 3.0, size=(n - nc, 3))
    perm = rng.permutation(n)
    pos = pos[perm].astype(np.float32)
    return _pack(pos, *_attributes(rng, n, scale_mul))


def overlap_stress_cloud(n, seed=4):
    """cfg 5: 90 % in 4 tight clusters (sigma 0.08), scales x4."""
    return clustered_cloud(n, seed=seed, n_clusters=4, sigma=0.08, frac=0.9, centre_extent=1.5, scale_mul=4.0)


def lattice_cloud(side, seed=3, spacing=0.5):
    """Tie-heavy set: positions on a cubic lattice symmetric about the origin, so a
    camera on a symmetry axis sees many exactly equal distances."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = (np.arange(side, dtype=np.float64) - (side - 1) / 2.0) * spacing
    pos = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    pos = pos[rng.permutation(pos.shape[0])].astype(np.float32)
    return _pack(pos, *_attributes(rng, pos.shape[0]))


SCENES = {"uniform": uniform_cloud, "clustered": clustered_cloud, "overlap": overlap_stress_cloud}
