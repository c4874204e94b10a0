"""gsx_sort_noalloc (include/gsx.h): the depth sort with its scratch in a file-scope device array, for callers that
cannot allocate.  Checked through the C ABI against gsx_sort -- same selection, same camera -> the same permutation,
byte for byte -- on two calls in a row, the second with a smaller input (the shared scratch keeps the first call's
contents), and against the oracle's contract order."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hip():
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    return hip


def test_sort_noalloc_matches_gsx_sort(gsx_fresh, pkg, oracle, synth):
    gsx = gsx_fresh
    hip = _hip()
    cap = 300_000
    out_dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(out_dev), cap * 4 + 64) == 0          # the CALLER's buffer (+ a guard tail)
    cams = [np.array(c, np.float32) for c in ([0.0, 0.0, 5.0], [1.7, -0.4, 4.2])]

    def run(n, seed, select=None):
        splats = oracle.raw_to_splat(synth.clustered_cloud(n, seed=seed))
        gsx.upload(splats)
        if select is not None:
            gsx.select_list(select)
        m = n if select is None else len(select)
        for cam in cams:
            want = gsx.sort(cam)                                        # the existing entry point
            assert len(want) == m
            assert hip.hipMemset(out_dev, 0xEE, cap * 4 + 64) == 0
            gsx.sort_noalloc(cam, out_dev.value, cap)                   # queued on the context's stream
            gsx.sync()
            got = np.empty(m + 16, np.uint32)
            assert hip.hipMemcpy(got.ctypes.data, out_dev.value, (m + 16) * 4, 2) == 0
            assert np.array_equal(got[:m], want)                        # bit-exact with gsx_sort
            assert np.all(got[m:] == 0xEEEEEEEE)                        # nothing written past the selection
            assert np.array_equal(want, oracle.sort_contract(splats, cam, select))
            dev, mm = gsx.sorted_dev()                                  # the context's own permutation is untouched
            own = np.empty(mm, np.uint32)
            assert hip.hipMemcpy(own.ctypes.data, dev, mm * 4, 2) == 0
            assert np.array_equal(own, want)

    run(250_000, seed=3)                                                # first call: large
    run(9_000, seed=4)                                                  # second call: smaller, stale scratch behind it
    run(60_000, seed=5, select=np.arange(1, 60_000, 3, dtype=np.uint32))   # a LOD-style sub-selection
    # empty selection: nothing written, no error
    gsx.select_list(np.zeros(0, np.uint32))
    assert hip.hipMemset(out_dev, 0xEE, 64) == 0
    gsx.sort_noalloc(cams[0], out_dev.value, cap)
    gsx.sync()
    got = np.empty(16, np.uint32)
    assert hip.hipMemcpy(got.ctypes.data, out_dev.value, 64, 2) == 0
    assert np.all(got == 0xEEEEEEEE)
    # capacity and the size limit are checked before anything is queued
    gsx.select_all()
    with pytest.raises(pkg.GsxError) as e:
        gsx.sort_noalloc(cams[0], out_dev.value, 100)
    assert e.value.code == pkg.GSX_E_INVALID
    big = np.zeros(((1 << 20) + 1, 20), np.float32)
    big[:, 12] = np.linspace(-1, 1, len(big), dtype=np.float32)
    big[:, 15] = 1.0
    big[:, 0] = big[:, 5] = big[:, 10] = 0.01
    gsx.upload(big)
    with pytest.raises(pkg.GsxError) as e:
        gsx.sort_noalloc(cams[0], out_dev.value, len(big))
    assert e.value.code == pkg.GSX_E_UNSUPPORTED and "exceeds" in str(e.value)
    assert hip.hipFree(out_dev) == 0
