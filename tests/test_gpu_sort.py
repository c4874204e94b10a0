"""GPU parity: upload, depth keys, the stable LSD radix sort and the LOD cuts, called through
the C ABI (libgsx.so) and compared bit for bit with the CPU oracle / golden fixtures."""
import numpy as np
import pytest

from conftest import bits, golden

pytestmark = pytest.mark.gpu

CAMS = [np.array(c, np.float32) for c in ([0.0, 0.0, 5.0], [1.7, -0.4, 4.2], [-6.0, 2.0, 0.5])]


def check_contract(perm, keys_by_id, m_ids):
    """perm is a permutation of m_ids ordered by key descending, index ascending."""
    assert np.array_equal(np.sort(perm), np.sort(m_ids))
    k = keys_by_id[perm]
    assert np.all(k[:-1] >= k[1:])
    same = k[1:] == k[:-1]
    assert np.all(perm[1:][same] > perm[:-1][same])


ALGOS = [1, 2]      # gsx_set_option("sort_algo"): 1 = the three LSD radix passes, 2 = the two-pass form (partition + in-LDS bucket sort) at any size


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4095, 4096, 4097, 8192, 50000, 262144 + 17])
def test_sort_matches_oracle(gsx_fresh, oracle, synth, n, algo):
    gsx = gsx_fresh
    splats = oracle.raw_to_splat(synth.uniform_cloud(n, seed=n % 97 + 1))
    gsx.set_option("sort_algo", algo)
    gsx.upload(splats)
    for cam in CAMS[:2]:
        keys = gsx.keys(cam)
        assert np.array_equal(bits(keys), bits(oracle.distance_keys(splats, cam)))   # float keys identical on CPU and GPU
        perm = gsx.sort(cam)
        assert np.array_equal(perm, oracle.sort_contract(splats, cam))               # bit-exact permutation


def test_sort_empty_scene(gsx, oracle):
    gsx.upload(np.zeros((0, 20), np.float32))
    assert len(gsx.sort(CAMS[0])) == 0


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("name", ["uniform1k", "lattice12", "clustered20k"])
def test_sort_golden(gsx_fresh, oracle, synth, name, algo):
    gsx = gsx_fresh
    g = golden("sort.npz")
    gsx.set_option("sort_algo", algo)
    raw = g[f"{name}_raw"] if f"{name}_raw" in g.files else synth.clustered_cloud(20000, seed=6)
    splats = oracle.raw_to_splat(raw)
    gsx.upload(splats)
    for p in range(3):
        cam = g[f"{name}_p{p}_cam"]
        ref_keys, ref_perm = g[f"{name}_p{p}_keys"], g[f"{name}_p{p}_perm"]
        assert np.array_equal(bits(gsx.keys(cam)), bits(ref_keys))
        perm = gsx.sort(cam)
        # vs the reference's (unstable) std::sort: same key sequence, same index set per equal-key run
        assert np.array_equal(bits(ref_keys[perm]), bits(ref_keys[ref_perm]))
        check_contract(perm, ref_keys, np.arange(len(ref_keys), dtype=np.uint32))
        assert np.array_equal(perm, oracle.sort_contract(splats, cam))


def test_sort_ties_all_equal_and_skipped_passes(gsx, oracle):
    # every splat at the same distance: all four digits constant -> passes 1..3 skipped, order = ascending index
    n = 10000
    splats = np.zeros((n, 20), np.float32)
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    splats[:, 12] = np.float32(2.0) * np.cos(ang).astype(np.float32)
    splats[:, 13] = np.float32(2.0) * np.sin(ang).astype(np.float32)
    splats[:, 15] = 1.0
    splats[:, 0] = splats[:, 5] = splats[:, 10] = 0.01
    gsx.upload(splats)
    cam = np.zeros(3, np.float32)
    perm = gsx.sort(cam)
    assert np.array_equal(perm, oracle.sort_contract(splats, cam))
    splats[:, 12:15] = [1.0, 2.0, 2.0]     # exactly equal distance 3.0 from the origin
    gsx.upload(splats)
    perm = gsx.sort(cam)
    assert np.array_equal(perm, np.arange(n, dtype=np.uint32))
    assert gsx.timings()["sort_passes"] == 1


@pytest.mark.parametrize("algo", ALGOS)
def test_sort_wide_key_range(gsx_fresh, oracle, algo):
    gsx = gsx_fresh
    # distances spanning many exponents (all four radix digits vary), zeros and huge values
    gsx.set_option("sort_algo", algo)
    rng = np.random.Generator(np.random.PCG64(5))
    n = 70000
    splats = np.zeros((n, 20), np.float32)
    mag = np.exp(rng.uniform(np.log(1e-18), np.log(1e18), size=n))
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    splats[:, 12:15] = (d * mag[:, None]).astype(np.float32)
    splats[:50, 12:15] = 0.0               # distance exactly 0 from a camera at the origin
    splats[:, 15] = 1.0
    gsx.upload(splats)
    cam = np.zeros(3, np.float32)
    assert np.array_equal(bits(gsx.keys(cam)), bits(oracle.distance_keys(splats, cam)))
    assert np.array_equal(gsx.sort(cam), oracle.sort_contract(splats, cam))
    assert gsx.timings()["sort_passes"] == (4 if algo == 1 else 2)     # (the two-pass form orders all 31 bits: its buckets widen towards the camera)


@pytest.mark.parametrize("algo", ALGOS)
def test_sort_camera_inside_scene_box(gsx_fresh, oracle, synth, algo):
    gsx = gsx_fresh
    """A camera inside the scene's bounding box: the host can only bound the nearest distance by 0, so the key range is
    the full 31 bits.  The low 27 bits take three passes; the top digit is a gated pass that runs only when the key
    kernel met a key that needs it (decided on the device).  Both outcomes must give the oracle's permutation."""
    n = 200_000
    base = 3 if algo == 1 else 2                                      # passes over the low 27 bits
    gsx.set_option("sort_algo", algo)
    splats = oracle.raw_to_splat(synth.clustered_cloud(n, seed=9))
    lo, hi = splats[:, 12:15].min(axis=0), splats[:, 12:15].max(axis=0)
    cam = (lo + (hi - lo) * np.array([0.45, 0.55, 0.5], np.float32)).astype(np.float32)
    d = oracle.distance_keys(splats, cam)
    assert d.min() > 1e-3 * d.max()                                   # nobody close enough to need the top digit
    gsx.upload(splats)
    assert np.array_equal(gsx.sort(cam), oracle.sort_contract(splats, cam))
    assert gsx.timings()["sort_passes"] == base                       # gated pass queued, found nothing to do
    near = splats.copy()                                              # three splats at 0 .. 1e-6 from the camera
    near[[5, 77_777, n - 1], 12:15] = cam + np.array([[1e-6, 0, 0], [0, 0, 0], [0, -3e-7, 2e-7]], np.float32)
    gsx.upload(near)
    assert np.array_equal(bits(gsx.keys(cam)), bits(oracle.distance_keys(near, cam)))
    assert np.array_equal(gsx.sort(cam), oracle.sort_contract(near, cam))
    assert gsx.timings()["sort_passes"] == (base + 1 if algo == 1 else base)   # top digit executed (the two-pass form has none)
    out = (hi + (hi - lo)).astype(np.float32)                         # outside the box again: the gate must be off, not stale
    assert np.array_equal(gsx.sort(out), oracle.sort_contract(near, out))
    assert gsx.timings()["sort_passes"] == base
    assert np.array_equal(gsx.sort(cam), oracle.sort_contract(near, cam))
    gsx.upload(splats)                                                # and a frame after a wide one starts from a cleared gate
    assert np.array_equal(gsx.sort(cam), oracle.sort_contract(splats, cam))
    assert gsx.timings()["sort_passes"] == base


def test_two_pass_dense_buckets_and_ties(gsx_fresh, oracle, synth):
    gsx = gsx_fresh
    """The two-pass form where its buckets overflow LDS (k2_big: one workgroup per oversized bucket, LSD over (key, index)),
    where every key of a bucket is equal (the tie rule inside one bin), and its automatic fall-back to the LSD passes."""
    n = 400_000
    rng = np.random.Generator(np.random.PCG64(11))
    raw = synth.uniform_cloud(n, seed=3)
    splats = oracle.raw_to_splat(raw)
    # 70 % of the splats in one blob of sigma 0.004: ~50 buckets of the partition hold 5000+ keys each, many keys collide
    k = int(0.7 * n)
    splats[:k, 12:15] = (np.array([0.3, -0.2, 0.1]) + 0.004 * rng.standard_normal((k, 3))).astype(np.float32)
    splats[1000:31000, 12:15] = np.float32([0.5, 0.5, -0.25])          # 30 000 splats at one point: one distance, 30 000 ties
    gsx.upload(splats)
    # automatic choice: the first sort takes the two-pass form, finds most keys in oversized buckets, and the following
    # ones run the LSD passes (same permutation either way)
    seen = []
    for i in range(4):
        assert np.array_equal(gsx.sort(CAMS[i % 3]), oracle.sort_contract(splats, CAMS[i % 3]))
        seen.append(gsx.timings()["sort_passes"])
    assert seen[0] == 2 and seen[-1] == 3, seen
    gsx.set_option("sort_algo", 2)
    for cam in CAMS:
        assert np.array_equal(gsx.sort(cam), oracle.sort_contract(splats, cam))
        assert gsx.timings()["sort_passes"] == 2


def test_two_pass_on_a_selection_list(gsx_fresh, oracle, synth):
    gsx = gsx_fresh
    n = 300_000
    splats = oracle.raw_to_splat(synth.clustered_cloud(n, seed=4))
    gsx.set_option("sort_algo", 2)
    gsx.upload(splats)
    ids = np.flatnonzero(np.random.Generator(np.random.PCG64(2)).random(n) < 0.6).astype(np.uint32)
    gsx.select_list(ids)
    for cam in CAMS[:2]:
        assert np.array_equal(gsx.sort(cam), oracle.sort_contract(splats, cam, ids))
        assert gsx.timings()["sort_passes"] == 2
    # an empty selection and a single splat go through the same five launches
    gsx.select_list(np.zeros(0, np.uint32))
    assert len(gsx.sort(CAMS[0])) == 0
    gsx.select_list(np.array([n - 1], np.uint32))
    assert np.array_equal(gsx.sort(CAMS[1]), np.array([n - 1], np.uint32))
    gsx.select_list(ids[:70])                                  # (a bucket of at most 64 pairs is ranked by one wave)
    assert np.array_equal(gsx.sort(CAMS[1]), oracle.sort_contract(splats, CAMS[1], ids[:70]))
    gsx.select_all()
    assert np.array_equal(gsx.sort(CAMS[0]), oracle.sort_contract(splats, CAMS[0]))


def test_two_pass_bucket_count_follows_the_cut_size(gsx_fresh, oracle, synth):
    """The two-pass sort sizes its buckets by the selection size recent sorts saw (a device-side cut's size never reaches the
    host otherwise).  A cut that jumps from one node to 262 144 nodes lying in a thin depth shell meets a plan sized for the
    small cut: buckets overflow LDS, the slow path sorts them, the hint is dropped - and every permutation is still exact."""
    gsx = gsx_fresh
    levels = 7
    starts = np.cumsum([0] + [8 ** l for l in range(levels)])
    n = int(starts[-1])                                          # 299 593 nodes, 262 144 of them leaves
    parent = np.full(n, 0xFFFFFFFF, np.uint32)
    for l in range(1, levels):
        j = np.arange(8 ** l)
        parent[starts[l] + j] = starts[l - 1] + j // 8
    leaf = np.zeros(n, np.uint8)
    leaf[starts[levels - 1]:] = 1
    rng = np.random.Generator(np.random.PCG64(11))
    splats = oracle.raw_to_splat(synth.uniform_cloud(n, seed=21)).copy()
    flat = splats.reshape(n, -1)
    assert flat.shape[1] == 20
    cam = np.array([0.3, -0.2, 0.1], np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    radius = (4.0 + 1.0e-4 * rng.random(n)).astype(np.float32)   # a shell 0.1 mm thick: a few hundred distinct keys, many ties
    flat[:, 12:15] = cam + d * radius[:, None]
    center3 = np.zeros((n, 3), np.float32)
    half3 = np.full((n, 3), 50.0, np.float32)
    gsx.set_option("sort_algo", 2)
    gsx.upload(splats)
    gsx.set_lod_octree(center3, half3, parent, leaf)
    vp = np.eye(4, dtype=np.float32).reshape(-1)
    leaves = np.arange(starts[levels - 1], n, dtype=np.uint32)
    big = oracle.sort_contract(splats, cam, leaves)
    one = oracle.sort_contract(splats, cam, np.zeros(1, np.uint32))
    for rep in range(2):
        for k in range(9):                                       # the root stops: one node (statistics reach the host every 4th sort)
            assert gsx.select_octree(vp, 1.0e30) == 1
            assert np.array_equal(gsx.sort(cam), one)
        for k in range(9):                                       # nothing stops before the leaves
            gsx.select_octree(vp, -1.0, want_count=False)
            assert np.array_equal(gsx.sort(cam), big), f"round {rep}, sort {k} after the jump"
            assert gsx.timings()["sort_form"] == 2


def test_sort_is_repeatable_and_view_dependent(gsx, oracle, synth):
    splats = oracle.raw_to_splat(synth.clustered_cloud(30000, seed=8))
    gsx.upload(splats)
    a = gsx.sort(CAMS[1])
    b = gsx.sort(CAMS[2])
    c = gsx.sort(CAMS[1])
    assert np.array_equal(a, c) and not np.array_equal(a, b)


def test_select_list_sorts_sublist(gsx, oracle, synth):
    splats = oracle.raw_to_splat(synth.uniform_cloud(20000, seed=12))
    gsx.upload(splats)
    rng = np.random.Generator(np.random.PCG64(1))
    sub = rng.permutation(20000)[:7777].astype(np.uint32)     # arbitrary (queue) order, like HC cuts
    gsx.select_list(sub)
    assert np.array_equal(gsx.selection(), np.sort(sub))
    for cam in CAMS:
        perm = gsx.sort(cam)
        assert np.array_equal(perm, oracle.sort_contract(splats, cam, np.sort(sub)))
    gsx.select_list(np.zeros(0, np.uint32))
    assert len(gsx.sort(CAMS[0])) == 0
    gsx.select_all()
    assert np.array_equal(gsx.sort(CAMS[0]), oracle.sort_contract(splats, CAMS[0]))


def _octree_parent(first_child, n_children):
    n = len(n_children)
    parent = np.full(n, 0xFFFFFFFF, np.uint32)
    for i in range(n):
        parent[first_child[i]:first_child[i] + n_children[i]] = i
    return parent


def test_octree_select_golden(gsx, oracle):
    g = golden("lod.npz")
    fc, nc = g["oc_first_child"], g["oc_n_children"]
    gsx.upload(g["oc_splats"])
    gsx.set_lod_octree(g["oc_center3"], g["oc_half3"], _octree_parent(fc, nc), (nc == 0).astype(np.uint8))
    for p in range(3):
        vp = g["pose_viewproj"][p]
        cam = g["pose_world"][p][12:15]
        for t, thr in enumerate(g["oc_thr"]):
            ref_cut = g[f"oc_cut_p{p}_t{t}"]
            m = gsx.select_octree(vp, float(thr))
            assert m == len(ref_cut)
            assert np.array_equal(gsx.selection(), ref_cut)          # Octree.cpp:189-216 (BFS order is ascending)
            perm = gsx.sort(cam)
            assert np.array_equal(perm, oracle.sort_contract(g["oc_splats"], cam, ref_cut))


def test_cut_without_count_then_sort_with_download(gsx_fresh, oracle):
    """A cut queued without asking for its size (it runs on the sort stream) followed at once by a sort that downloads the
    permutation: the size read back must be this cut's, not the previous one's."""
    gsx = gsx_fresh
    g = golden("lod.npz")
    fc, nc = g["oc_first_child"], g["oc_n_children"]
    gsx.upload(g["oc_splats"])
    gsx.set_lod_octree(g["oc_center3"], g["oc_half3"], _octree_parent(fc, nc), (nc == 0).astype(np.uint8))
    sizes = set()
    for rep in range(3):
        for p in range(3):
            cam = g["pose_world"][p][12:15]
            for t, thr in enumerate(g["oc_thr"]):
                ref_cut = g[f"oc_cut_p{p}_t{t}"]
                sizes.add(len(ref_cut))
                gsx.select_octree(g["pose_viewproj"][p], float(thr), want_count=False)
                perm = gsx.sort(cam)
                assert len(perm) == len(ref_cut)
                assert np.array_equal(perm, oracle.sort_contract(g["oc_splats"], cam, ref_cut))
                dev, m = gsx.sorted_dev()
                assert m == len(ref_cut)
    assert len(sizes) > 1


def _hc_parent(child2, n):
    parent = np.full(n, 0xFFFFFFFF, np.uint32)
    c = child2.reshape(-1, 2)
    for i in range(n):
        if c[i, 0] != 0xFFFFFFFF:
            parent[c[i, 0]] = i
            parent[c[i, 1]] = i
    return parent


@pytest.mark.parametrize("prefix,thr_key", [("hc", "hc_mthr"), ("g", "g_mthr")])
def test_hc_metric_select_golden(gsx, oracle, prefix, thr_key):
    g = golden("lod.npz")
    splats, child2 = g[f"{prefix}_splats"], g[f"{prefix}_child2"]
    n = len(splats)
    leaf = (child2.reshape(-1, 2)[:, 0] == 0xFFFFFFFF).astype(np.uint8)
    gsx.upload(splats)
    gsx.set_lod_hc(_hc_parent(child2, n), leaf, g[f"{prefix}_error"], g[f"{prefix}_weight"])
    for p in range(3):
        cam = g["pose_world"][p][12:15]
        for t, thr in enumerate(g[thr_key]):
            ref_cut = g[f"{prefix}_cut_metric_p{p}_t{t}"]
            m = gsx.select_hc(cam, float(thr))
            assert m == len(ref_cut)
            assert np.array_equal(gsx.selection(), np.sort(ref_cut))  # same index set (reference: queue order)
            assert np.array_equal(gsx.sort(cam), oracle.sort_contract(splats, cam, np.sort(ref_cut)))


def test_gridhc_depth_cut_through_select_list(gsx, oracle):
    g = golden("lod.npz")
    gsx.upload(g["g_splats"])
    cam = g["pose_world"][1][12:15]
    for d in g["g_depths"]:
        ref_cut = g[f"g_cut_depth_{d}"]
        gsx.select_list(ref_cut)
        assert np.array_equal(gsx.sort(cam), oracle.sort_contract(g["g_splats"], cam, np.sort(ref_cut)))


def test_gridhc_metric_cut_at_250k(gsx_fresh, pkg, oracle, synth):
    """cfg 3's other LOD structure beyond fixture size: GridHC (20^3 cells, GridHC.cpp:10-74) over 250 K clustered splats,
    built by the host builder (bit-identical to the reference's, tests/test_host.py), cut on the device by the error metric
    (HC.cpp:100-132 per cell root) along the orbit, against the CPU oracle's cut of every cell; the permutation of one cut
    against the sort contract."""
    import importlib
    lod = importlib.import_module("gaussian_splatting_amd.lod")
    gsx = gsx_fresh
    splats = oracle.raw_to_splat(synth.clustered_cloud(250_000, seed=9))
    t = lod.build_gridhc(splats)
    assert t.n > 400_000 and len(t.roots) > 1000
    t.upload(gsx)
    crs = t.cell_root_start
    pos = np.ascontiguousarray(t.splats[:, 12:15])
    err = t.error * t.weight
    thrs = [float(np.quantile(err[err > 0], q)) / 25.0 for q in (0.5, 0.9, 0.99)]
    sizes = []
    for f in (0, 40, 80):
        world, _ = oracle.orbit_world(5.0, (0.0, 1.0, 0.0), 2.0 * np.pi / 120.0, f)
        cam = np.ascontiguousarray(world[12:15])
        for thr in thrs:
            want = np.sort(np.concatenate([oracle.hc_cut_metric(np.ascontiguousarray(t.roots[crs[c]:crs[c + 1]]), t.child2, t.error, t.weight, pos, cam, thr)
                                           for c in range(len(crs) - 1) if crs[c + 1] > crs[c]]))
            assert gsx.select_hc(cam, thr) == len(want)
            assert np.array_equal(gsx.selection(), want)
            sizes.append(len(want))
    assert min(sizes) < 0.8 * max(sizes) and min(sizes) >= len(t.roots)     # the thresholds cut at different depths
    perm = gsx.sort(cam)
    assert np.array_equal(perm, oracle.sort_contract(t.splats, cam, want))
    t.close()


def test_error_flags_are_reported_once(gsx_fresh, pkg, oracle, synth):
    """A bad selection list (id >= n) or a non-finite position is reported by the call that meets it and by no later one:
    the device error words are cleared once read, so the context keeps working (they used to be sticky)."""
    gsx = gsx_fresh
    n, w, h = 5000, 96, 64
    splats = oracle.raw_to_splat(synth.uniform_cloud(n, seed=3))
    gsx.upload(splats)
    cam = CAMS[0]
    bad = np.array([1, 2, n + 5, 7], np.uint32)
    with pytest.raises(pkg.GsxError) as e:
        gsx.select_list(bad)
    assert e.value.code == pkg.GSX_E_INVALID
    good = np.arange(0, n, 2, dtype=np.uint32)
    gsx.select_list(good)                                             # a valid list right after: no stale error
    perm = gsx.sort(cam)
    assert np.array_equal(perm, oracle.sort_contract(splats, cam, good))
    world, _ = oracle.orbit_world(5.0, (0.0, 1.0, 0.0), 0.0, 0)
    view, proj, vp, campos = oracle.camera(world, 45.0, w / h)
    img = gsx.render(pkg.make_uniforms(proj, view), w, h)
    assert img[..., :3].std() > 1
    gsx.sync()
    # a NaN position leaves the host's distance bound: the sort that meets it reports GSX_E_INTERNAL, the next upload is clean
    broken = splats.copy()
    broken[17, 12] = np.nan
    gsx.upload(broken)
    with pytest.raises(pkg.GsxError) as e:
        gsx.sort(cam)
    assert e.value.code == pkg.GSX_E_INTERNAL
    gsx.select_list(good)                                             # no stale key_bound error on an unrelated call
    gsx.upload(splats)
    assert np.array_equal(gsx.sort(cam), oracle.sort_contract(splats, cam))


def test_hc_forest_with_a_cycle_is_rejected(gsx_fresh, pkg, oracle, synth):
    gsx = gsx_fresh
    n = 64
    gsx.upload(oracle.raw_to_splat(synth.uniform_cloud(n, seed=2)))
    parent = np.full(n, 0xFFFFFFFF, np.uint32)
    parent[:10] = np.arange(1, 11)            # a chain 0 -> 1 -> ... -> 10 (root): fine
    leaf = np.ones(n, np.uint8)
    gsx.set_lod_hc(parent, leaf, np.ones(n, np.float32), np.ones(n, np.float32))
    parent[10] = 3                            # 3 -> 4 -> ... -> 10 -> 3
    with pytest.raises(pkg.GsxError) as e:
        gsx.set_lod_hc(parent, leaf, np.ones(n, np.float32), np.ones(n, np.float32))
    assert e.value.code == pkg.GSX_E_INVALID
    parent[:] = 0xFFFFFFFF
    parent[5] = 5                             # self loop
    with pytest.raises(pkg.GsxError):
        gsx.set_lod_hc(parent, leaf, np.ones(n, np.float32), np.ones(n, np.float32))


def test_cfg1_full_size_bit_exact(gsx, oracle, synth):
    """BASELINE.json configs[0] at its full size: 1M-splat uniform cloud, default OrbitCamera(5) pose, keys + sort,
    byte for byte against the oracle's contract order."""
    n = 1_000_000
    splats = oracle.raw_to_splat(synth.uniform_cloud(n, seed=1))
    gsx.upload(splats)
    cam = np.array([0.0, 0.0, 5.0], np.float32)
    assert np.array_equal(bits(gsx.keys(cam)), bits(oracle.distance_keys(splats, cam)))
    assert np.array_equal(gsx.sort(cam), oracle.sort_contract(splats, cam))


def test_upload_rejects_asymmetric_covariance(gsx, pkg):
    s = np.zeros((4, 20), np.float32)
    s[:, 0] = s[:, 5] = s[:, 10] = 0.01
    s[:, 15] = 1.0
    s[2, 1] = 0.5           # transform[0][1] != transform[1][0]
    with pytest.raises(pkg.GsxError) as e:
        gsx.upload(s)
    assert e.value.code == pkg.GSX_E_UNSUPPORTED


def test_large_sort_properties(gsx, oracle, synth):
    """cfg-2 size (6M): size-independent properties instead of a full oracle sort."""
    n = 6_000_000
    raw = synth.clustered_cloud(n, seed=6)
    splats = oracle.raw_to_splat(raw)
    gsx.upload(splats)
    cam = np.array([0.0, 0.0, 5.0], np.float32)
    perm = gsx.sort(cam)
    keys = oracle.distance_keys(splats, cam)
    k = keys[perm]
    assert np.all(k[:-1] >= k[1:])                                   # sortedness
    seen = np.zeros(n, bool)
    seen[perm] = True
    assert seen.all() and len(perm) == n                             # a permutation
    same = k[1:] == k[:-1]
    assert np.all(perm[1:][same] > perm[:-1][same])                  # ties by ascending index
    # idempotence: sorting the already-sorted selection order gives the same permutation
    assert np.array_equal(gsx.sort(cam), perm)
    # spot check a window against the oracle's full order
    sub = np.sort(perm[1000:21000])
    gsx.select_list(sub)
    assert np.array_equal(gsx.sort(cam), oracle.sort_contract(splats, cam, sub))
    gsx.select_all()


def test_octree_select_large_orbit(gsx, pkg, oracle, synth):
    """cfg-3 shape at 1M splats (1.45M octree nodes): the device cut equals the CPU restatement of Octree::get_indices
    along an orbit, and the LOD -> sort chain stays on the device."""
    import importlib
    lod = importlib.import_module("gaussian_splatting_amd.lod")
    t = lod.build_octree(synth.clustered_cloud(1_000_000, seed=6))    # host builder, bit-exact with the reference (test_host.py)
    t.upload(gsx)
    for steps in (0, 40, 80):                                         # frames of a 120-frame orbit
        world, _ = oracle.orbit_world(5.0, (0.0, 1.0, 0.0), 2.0 * np.pi / 120.0, steps)
        view, proj, vp, cam = oracle.camera(world, 45.0, 16.0 / 9.0)
        for thr in (0.002, 0.0002):                                   # min_screen_area 0.2 * 0.01 (SplatMeshOctree.hpp:44) and a finer cut
            ref_cut = oracle.octree_cut(t.center3, t.half3, t.first_child, t.n_children, vp, np.float32(thr))
            m = gsx.select_octree(vp, float(thr))
            assert m == len(ref_cut) and 0 < m < t.n
            assert np.array_equal(gsx.selection(), ref_cut)
        perm = gsx.sort(cam)
        k = oracle.distance_keys(t.splats, cam)[perm]
        assert np.all(k[:-1] >= k[1:]) and np.array_equal(np.sort(perm), ref_cut)
    gsx.select_all()
    t.close()


def test_octree_non_bfs_layout_uses_the_general_cut(gsx_fresh, oracle):
    """The level-by-level cut needs BFS order.  The golden tree renumbered in depth-first preorder (parents still precede
    children, levels no longer contiguous) must give the same SET, through the ancestor-walk kernels."""
    gsx = gsx_fresh
    g = golden("lod.npz")
    fc, nc = g["oc_first_child"], g["oc_n_children"]
    parent = _octree_parent(fc, nc)
    n = len(parent)
    order, stack = [], [0]
    while stack:                                          # preorder: order[k] = old id of the node numbered k
        i = stack.pop()
        order.append(i)
        stack.extend(range(int(fc[i]) + int(nc[i]) - 1, int(fc[i]) - 1, -1))
    order = np.array(order)
    assert len(order) == n
    new_id = np.empty(n, np.int64)
    new_id[order] = np.arange(n)
    p2 = np.full(n, 0xFFFFFFFF, np.uint32)
    p2[new_id[1:]] = new_id[parent[1:]]
    assert np.all(p2[1:] < np.arange(1, n)) and np.any(np.diff(p2[1:].astype(np.int64)) < 0)    # topological, not BFS
    splats = np.zeros((n, 20), np.float32)
    splats[:, 0] = splats[:, 5] = splats[:, 10] = 0.01
    splats[:, 15] = 1.0
    gsx.upload(splats)
    gsx.set_lod_octree(g["oc_center3"][order], g["oc_half3"][order], p2, (nc == 0).astype(np.uint8)[order])
    for p in range(3):
        vp = g["pose_viewproj"][p]
        for t, thr in enumerate(g["oc_thr"]):
            m = gsx.select_octree(vp, float(thr))
            want = np.sort(new_id[g[f"oc_cut_p{p}_t{t}"]]).astype(np.uint32)
            assert m == len(want) and np.array_equal(gsx.selection(), want)


def _random_bfs_tree(rng, levels, first_width, grow, leaf_everywhere=True):
    """A BFS-numbered tree with nested boxes: level l has about first_width * grow^l nodes, every node 0..8 children."""
    parent, center, half, lvl = [0xFFFFFFFF], [np.zeros(3)], [np.full(3, 4.0)], [0]
    begin, end = 0, 1
    for l in range(1, levels):
        want = max(1, int(first_width * grow ** l))
        kids = np.zeros(end - begin, int)
        for _ in range(want):
            kids[rng.integers(0, end - begin)] += 1
        kids = np.minimum(kids, 8)
        if kids.sum() == 0:
            kids[0] = 1
        for j, k in enumerate(kids):
            p = begin + j
            for q in range(k):
                off = np.array([(q & 1) - 0.5, ((q >> 1) & 1) - 0.5, ((q >> 2) & 1) - 0.5]) * half[p]
                parent.append(p); center.append(center[p] + off * rng.uniform(0.6, 1.0)); half.append(half[p] * rng.uniform(0.35, 0.55, 3)); lvl.append(l)
        begin, end = end, len(parent)
    parent = np.array(parent, np.uint32)
    n = len(parent)
    has_child = np.zeros(n, bool)
    has_child[parent[1:]] = True
    leaf = (~has_child).astype(np.uint8)
    if not leaf_everywhere:
        leaf[np.array(lvl) == levels - 1] = 0             # childless nodes that are not marked as leaves: emitted only when small enough
    return parent, np.array(center, np.float32), np.array(half, np.float32), leaf


@pytest.mark.parametrize("shape", ["one node", "two levels", "wide and shallow", "deep", "deepest level not leaves", "very deep"])
def test_octree_cut_shapes_match_the_ancestor_walk(gsx_fresh, oracle, shape):
    """The three-launch cut (head, tail, one-launch compaction) against the ancestor-walk kernels and the per-level launches on
    trees whose shape exercises every branch of the launch logic: no level below the top, one, two; more levels below the
    queue than the LDS stack is sized for (per-level launches first, then the tail); a deepest level that is not all
    leaves (its nodes are then tested like any other)."""
    gsx = gsx_fresh
    rng = np.random.default_rng(len(shape))
    levels, width, grow, leafy = {"one node": (1, 1, 1, True), "two levels": (2, 1, 8, True), "wide and shallow": (7, 1, 5.5, True),
                                  "deep": (12, 6, 2.2, True), "deepest level not leaves": (9, 4, 3.0, False),
                                  "very deep": (20, 40, 1.45, True)}[shape]
    parent, center, half, leaf = _random_bfs_tree(rng, levels, width, grow, leafy)
    n = len(parent)
    splats = np.zeros((n, 20), np.float32)
    splats[:, 0] = splats[:, 5] = splats[:, 10] = 0.01
    splats[:, 15] = 1.0
    splats[:, 16:19] = center
    gsx.upload(splats)
    gsx.set_lod_octree(center, half, parent, leaf)
    world, _ = oracle.orbit_world(9.0, (0.0, 1.0, 0.0), 0.3, 1)
    view, proj, vp, cam = oracle.camera(world, 45.0, 16 / 9)
    for thr in (10.0, 0.5, 0.05, 0.004, 0.0003, 0.00002, 0.0):
        gsx.set_option("lod_walk", 1)
        m_ref = gsx.select_octree(vp, thr)
        ref = gsx.selection()
        for walk in (2, 0, 0):                            # (twice: the second cut reuses the cleared counters and the next stamp)
            gsx.set_option("lod_walk", walk)
            assert gsx.select_octree(vp, thr) == m_ref, (shape, thr, walk)
            assert np.array_equal(gsx.selection(), ref), (shape, thr, walk)
    gsx.set_option("lod_walk", 0)


def test_octree_cut_thresholds_on_node_areas(gsx_fresh, oracle):
    """The walk decides `screen_area < threshold` from a cheap evaluation with an error bound and only evaluates the exactly
    rounded area when that is too close to call.  Thresholds placed ON the exact area of a node (the node must not stop),
    one ulp above (it must) and one below, for nodes of every level and boxes near and behind the camera plane: the cut
    must be the CPU oracle's (oracle/orc.c, Octree.cpp:189-216) every time."""
    gsx = gsx_fresh
    rng = np.random.default_rng(17)
    parent, center, half, leaf = _random_bfs_tree(rng, 8, 2, 3.2)
    n = len(parent)
    nc = np.bincount(parent[1:], minlength=n).astype(np.uint32)
    fc = np.zeros(n, np.uint32)
    kids = np.flatnonzero(nc)
    fc[kids] = np.searchsorted(parent[1:], kids).astype(np.uint32) + 1
    splats = np.zeros((n, 20), np.float32)
    splats[:, 0] = splats[:, 5] = splats[:, 10] = 0.01
    splats[:, 15] = 1.0
    splats[:, 16:19] = center
    gsx.upload(splats)
    gsx.set_lod_octree(center, half, parent, leaf)
    checked = 0
    for dist in (9.0, 2.5, 0.7):                          # (0.7: the camera stands inside the tree's box, corners behind it)
        world, _ = oracle.orbit_world(dist, (0.0, 1.0, 0.0), 0.4, 2)
        view, proj, vp, cam = oracle.camera(world, 45.0, 16 / 9)
        for i in rng.choice(np.flatnonzero(leaf == 0), 40, replace=False):
            area = np.float32(oracle.bb_screen_area(center[i], half[i], vp))
            if not np.isfinite(area) or area <= 0:
                continue
            for thr in (area, np.nextafter(area, np.float32(np.inf)), np.nextafter(area, np.float32(0))):
                want = oracle.octree_cut(center, half, fc, nc, vp, float(thr))
                assert gsx.select_octree(vp, float(thr)) == len(want), (dist, i, thr)
                assert np.array_equal(gsx.selection(), np.sort(want)), (dist, i, thr)
                checked += 1
    assert checked > 200


def test_one_launch_compaction_edges(gsx_fresh, oracle):
    """Selection lists through the one-launch compaction: sizes that are not multiples of 16 or of a tile, nothing selected,
    everything selected (more ids than a workgroup's flags are sparse), the last id alone; against the three launches."""
    gsx = gsx_fresh
    rng = np.random.default_rng(5)
    for n in (1, 15, 16, 17, 8191, 8193, 300_001, 2_500_003):
        splats = np.zeros((n, 20), np.float32)
        splats[:, 0] = splats[:, 5] = splats[:, 10] = 0.01
        splats[:, 15] = 1.0
        splats[:, 16:19] = rng.uniform(-1, 1, (n, 3))
        gsx.upload(splats)
        picks = [np.zeros(0, np.uint32), np.arange(n, dtype=np.uint32), np.array([n - 1], np.uint32),
                 np.flatnonzero(rng.random(n) < 0.13).astype(np.uint32), np.flatnonzero(rng.random(n) < 0.9).astype(np.uint32)]
        for pick in picks:
            for walk in (0, 1):
                gsx.set_option("lod_walk", walk)
                gsx.select_list(rng.permutation(pick))
                assert np.array_equal(gsx.selection(), pick), (n, len(pick), walk)
    gsx.set_option("lod_walk", 0)


def test_cfg3_octree_orbit_at_6m(gsx_fresh, pkg, oracle, synth):
    """cfg 3 at full size: 6 M splats -> 8.48 M octree nodes, frames of the 120-frame orbit.  The top-down level-by-level
    cut against (a) the general ancestor-walk kernels on every tested frame and (b) the host's Octree::get_indices
    (host/gsx_lod_host.hpp, bit-identical to the reference per tests/test_host.py) on three of them; then the
    LOD -> sort -> frame chain by properties."""
    import importlib
    lod = importlib.import_module("gaussian_splatting_amd.lod")
    gsx = gsx_fresh
    t = lod.build_octree(synth.clustered_cloud(6_000_000, seed=6))
    assert t.n > 8_000_000
    t.upload(gsx)
    w, h = 1920, 1080
    sizes = []
    for f in range(0, 120, 8):
        world, _ = oracle.orbit_world(5.0, (0.0, 1.0, 0.0), 2.0 * np.pi / 120.0, f)
        view, proj, vp, cam = oracle.camera(world, 45.0, w / h)
        gsx.set_option("lod_walk", 1)
        m_walk = gsx.select_octree(vp, 0.002)
        walk = gsx.selection()
        gsx.set_option("lod_walk", 2)                    # a launch per level, three-launch compaction (round 2's form)
        assert gsx.select_octree(vp, 0.002) == m_walk and np.array_equal(gsx.selection(), walk)
        gsx.set_option("lod_walk", 0)                    # head + tail + one-launch compaction
        m = gsx.select_octree(vp, 0.002)
        cut = gsx.selection()
        assert m == m_walk and np.array_equal(cut, walk)
        assert np.all(cut[1:] > cut[:-1]) and 100_000 < m < t.n
        sizes.append(m)
        if f % 40 == 0:
            ref = np.sort(t.cut_camera(world, 0.002, 45.0, w / h))
            assert np.array_equal(cut, ref)
            perm = gsx.sort(cam)
            assert np.array_equal(np.sort(perm), cut)                  # the sort permutes exactly the cut
            k = oracle.distance_keys(t.splats[perm], cam)
            assert np.all(k[:-1] >= k[1:])
            u = pkg.make_uniforms(proj, view)
            img = gsx.render(u, w, h)
            assert np.all(img[..., 3] == 255) and img[..., :3].std() > 10
            assert np.array_equal(gsx.render(u, w, h), img)
    assert len(set(sizes)) > 4                                          # the cut follows the camera
    # a finer and a coarser threshold, and the extremes: everything stops at the root / nothing stops before the leaves
    world, _ = oracle.orbit_world(5.0, (0.0, 1.0, 0.0), 0.3, 1)
    view, proj, vp, cam = oracle.camera(world, 45.0, w / h)
    for thr in (0.0002, 0.05):
        m = gsx.select_octree(vp, thr)
        assert np.array_equal(gsx.selection(), np.sort(t.cut_camera(world, thr, 45.0, w / h)))
    assert gsx.select_octree(vp, 1e9) == 1 and gsx.selection()[0] == 0
    m = gsx.select_octree(vp, 0.0)
    assert m == int(t.is_leaf.sum()) and np.array_equal(gsx.selection(), np.nonzero(t.is_leaf)[0].astype(np.uint32))
    # Two contexts on one GPU cutting at the same time, frames running beside the cuts: the one-launch compaction's workgroups
    # wait for each other's counts, and here they share the GPU with a second grid of the same kind and with rasterizer launches.
    vps, cuts = [], []
    for f in (3, 33, 63, 93):
        world, _ = oracle.orbit_world(5.0, (0.0, 1.0, 0.0), 2.0 * np.pi / 120.0, f)
        view, proj, vp, cam = oracle.camera(world, 45.0, w / h)
        gsx.select_octree(vp, 0.002)
        vps.append((vp, cam, pkg.make_uniforms(proj, view)))
        cuts.append(gsx.selection())
    with pkg.Gsx(0) as g2:
        t.upload(g2)
        for rep in range(6):
            for k in range(4):
                a, b = vps[k], vps[(k + rep + 1) % 4]
                gsx.select_octree(a[0], 0.002, want_count=False); gsx.sort(a[1], download=False); gsx.render(a[2], w, h, download=False)
                g2.select_octree(b[0], 0.002, want_count=False); g2.sort(b[1], download=False); g2.render(b[2], w, h, download=False)
                gsx.select_octree(b[0], 0.002, want_count=False)
                g2.select_octree(a[0], 0.002, want_count=False)
                assert np.array_equal(gsx.selection(), cuts[(k + rep + 1) % 4])
                assert np.array_equal(g2.selection(), cuts[k])
    gsx.select_all()
    t.close()
