"""CPU: the oracle restatement (oracle/orc_*.cpp) against fixtures produced by the
reference's own compiled sources (tests/golden/make_golden.py), bit for bit."""
import numpy as np
import pytest

from conftest import bits, golden


def test_datamodel_raw_to_split_and_splat(oracle):
    g = golden("datamodel.npz")
    split = oracle.raw_to_split(g["raw"])
    assert np.array_equal(bits(split), bits(g["split"]))          # Splat.h:60-75
    splat = oracle.split_to_splat(split)
    assert np.array_equal(bits(splat), bits(g["splat"]))          # Splat.h:77-91
    assert np.array_equal(bits(oracle.splat_weight(splat)), bits(g["weight"]))  # Splat.h:24-28


def test_splat_record_layout(oracle):
    g = golden("datamodel.npz")
    s = g["splat"]
    assert s.shape[1] * 4 == 80                                   # sizeof(Splat)
    assert np.all(s[:, 15] == 1.0) and np.all(s[:, [3, 7, 11]] == 0.0)
    # covariance block symmetric bit for bit (what gsx_upload_splats relies on)
    m = s[:, :16].reshape(-1, 4, 4)
    assert np.array_equal(bits(m[:, :3, :3]), bits(np.transpose(m[:, :3, :3], (0, 2, 1))))


def test_camera_orbit_matrices(oracle):
    g = golden("camera_orbit.npz")
    rot = None
    for s in range(g["world"].shape[0]):
        world, rot = oracle.orbit_world(5.0, (0.0, 1.0, 0.0), float(g["angle"]), 1 if s else 0, rot3=rot)
        view, proj, vp, cam = oracle.camera(world, 45.0, float(g["aspect"]))
        assert np.array_equal(bits(world), bits(g["world"][s])), s    # Node.cpp:10-27
        assert np.array_equal(bits(view), bits(g["view"][s])), s      # Camera.h:27-29
        assert np.array_equal(bits(proj), bits(g["proj"][s])), s      # Camera.h:32-34
        assert np.array_equal(bits(vp), bits(g["viewproj"][s])), s    # BB.hpp:60-61


def _sort_inputs(g, name, synth, oracle):
    if f"{name}_raw" in g.files:
        raw = g[f"{name}_raw"]
    else:
        raw = synth.clustered_cloud(20000, seed=6)
        assert int(raw.astype(np.uint64).sum()) == int(g[f"{name}_rawsum"][0]), "synthetic generator drifted"
    return oracle.raw_to_splat(raw)


@pytest.mark.parametrize("name", ["uniform1k", "lattice12", "clustered20k"])
def test_sort_keys_and_order(oracle, synth, name):
    g = golden("sort.npz")
    splats = _sort_inputs(g, name, synth, oracle)
    for p in range(3):
        cam = g[f"{name}_p{p}_cam"]
        keys = oracle.distance_keys(splats, cam)
        assert np.array_equal(bits(keys), bits(g[f"{name}_p{p}_keys"]))     # SplatMesh.h:90-94
        perm = oracle.sort_contract(splats, cam)
        ref_perm = g[f"{name}_p{p}_perm"]
        # the reference's std::sort is unstable: equal key SEQUENCE, equal index SET per equal-key run
        assert np.array_equal(bits(keys[perm]), bits(keys[ref_perm]))
        assert np.array_equal(np.sort(perm), np.arange(len(perm)))
        k = keys[perm]
        assert np.all(k[:-1] >= k[1:])
        run = np.concatenate([[0], np.cumsum(k[1:] != k[:-1])])
        order_o = np.lexsort((perm, run))
        order_r = np.lexsort((ref_perm, run))
        assert np.array_equal(perm[order_o], ref_perm[order_r])
        # contract tie rule: ascending index inside every equal-key run
        same = k[1:] == k[:-1]
        assert np.all(perm[1:][same] > perm[:-1][same])


def test_lattice_has_ties(oracle):
    g = golden("sort.npz")
    keys = g["lattice12_p0_keys"]
    assert len(np.unique(keys)) < len(keys) // 4


def test_sort_reference_port_matches_contract_keys(oracle, synth):
    splats = oracle.raw_to_splat(synth.uniform_cloud(5000, seed=2))
    cam = np.array([0.3, -0.2, 5.0], np.float32)
    idx = np.arange(5000, dtype=np.uint32)
    secs = oracle.sort_reference_port(splats, cam, idx)
    assert secs > 0
    keys = oracle.distance_keys(splats, cam)
    assert np.array_equal(bits(keys[idx]), bits(keys[oracle.sort_contract(splats, cam)]))


def test_sort_reference_port_parallel_matches_contract_keys(oracle, synth):
    """The multi-threaded baseline port (stand-in for the reference's PARALLEL build) yields the same key sequence."""
    n = 200_000                                        # above libstdc++'s parallel-mode threshold, so threads really run
    splats = oracle.raw_to_splat(synth.uniform_cloud(n, seed=2))
    cam = np.array([0.3, -0.2, 5.0], np.float32)
    idx = np.arange(n, dtype=np.uint32)
    secs = oracle.sort_reference_port_parallel(splats, cam, idx, threads=4)
    assert secs > 0 and np.array_equal(np.sort(idx), np.arange(n, dtype=np.uint32))
    keys = oracle.distance_keys(splats, cam)
    assert np.array_equal(bits(keys[idx]), bits(keys[oracle.sort_contract(splats, cam)]))


def test_sort_contract_sublist_and_empty(oracle, synth):
    splats = oracle.raw_to_splat(synth.uniform_cloud(300, seed=5))
    cam = np.array([0, 0, 5], np.float32)
    sub = np.array([7, 3, 250, 100, 101], np.uint32)
    out = oracle.sort_contract(splats, cam, sub)
    keys = oracle.distance_keys(splats, cam)
    assert sorted(out.tolist()) == sorted(sub.tolist())
    assert np.all(keys[out][:-1] >= keys[out][1:])
    assert len(oracle.sort_contract(splats, cam, np.zeros(0, np.uint32))) == 0


def test_octree_cut_and_screen_area(oracle):
    g = golden("lod.npz")
    c3, h3, fc, nc = g["oc_center3"], g["oc_half3"], g["oc_first_child"], g["oc_n_children"]
    for p in range(3):
        vp = g["pose_viewproj"][p]
        area = np.array([oracle.bb_screen_area(c3[i], h3[i], vp) for i in range(len(nc))], np.float32)
        assert np.array_equal(bits(area), bits(g[f"oc_area_p{p}"]))               # BB.hpp:59-77
        for t, thr in enumerate(g["oc_thr"]):
            cut = oracle.octree_cut(c3, h3, fc, nc, vp, float(thr))
            assert np.array_equal(cut, g[f"oc_cut_p{p}_t{t}"])                    # Octree.cpp:189-216
            assert np.all(np.diff(cut.astype(np.int64)) > 0)


def test_hc_cuts(oracle):
    g = golden("lod.npz")
    for d in g["hc_depths"]:
        assert np.array_equal(oracle.hc_cut_depth(g["hc_roots"], g["hc_child2"], int(d)), g[f"hc_cut_depth_{d}"])  # HC.cpp:134-168
    pos = np.ascontiguousarray(g["hc_splats"][:, 12:15])
    for p in range(3):
        cam = g["pose_world"][p][12:15]
        for t, thr in enumerate(g["hc_mthr"]):
            cut = oracle.hc_cut_metric(g["hc_roots"], g["hc_child2"], g["hc_error"], g["hc_weight"], pos, cam, float(thr))
            assert np.array_equal(cut, g[f"hc_cut_metric_p{p}_t{t}"])             # HC.cpp:100-132


def test_gridhc_cuts(oracle):
    g = golden("lod.npz")
    crs = g["g_cell_root_start"]
    cells = [np.ascontiguousarray(g["g_roots"][crs[c]:crs[c + 1]]) for c in range(len(crs) - 1)]
    for d in g["g_depths"]:
        cut = np.concatenate([oracle.hc_cut_depth(r, g["g_child2"], int(d)) for r in cells])
        assert np.array_equal(cut, g[f"g_cut_depth_{d}"])                         # GridHC.cpp:76-92
    pos = np.ascontiguousarray(g["g_splats"][:, 12:15])
    for p in range(3):
        cam = g["pose_world"][p][12:15]
        for t, thr in enumerate(g["g_mthr"]):
            cut = np.concatenate([oracle.hc_cut_metric(r, g["g_child2"], g["g_error"], g["g_weight"], pos, cam, float(thr))
                                  for r in cells])
            assert np.array_equal(cut, g[f"g_cut_metric_p{p}_t{t}"])              # GridHC.cpp:94-111


def test_oracle_matches_live_reference(oracle, reference, synth):
    """When oracle/_ref is present (build container and GPU box), re-pin on a fresh seed."""
    raw = synth.uniform_cloud(3000, seed=99)
    assert np.array_equal(bits(oracle.raw_to_split(raw)), bits(reference.raw_to_split(raw)))
    sp = oracle.raw_to_splat(raw)
    assert np.array_equal(bits(sp), bits(reference.split_to_splat(reference.raw_to_split(raw))))
    cam = np.array([1.5, 0.25, -4.0], np.float32)
    assert np.array_equal(bits(oracle.distance_keys(sp, cam)), bits(reference.distance_keys(sp, cam)))
