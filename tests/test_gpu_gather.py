"""gsx_gather through the C ABI on one GPU (include/gsx.h): a one-rank communicator gathers from itself with device
copies — the offset / in-place rules of the exchange, its ordering behind a queued frame, and argument checking.
The N > 1 legs (grouped RCCL send / recv) need more than one GPU: their argument contract is exercised at world size 2
in tests/test_distributed_cpu.py and the real thing by `bench.py --gpus N`, which checks the gather before trusting it."""
import ctypes as C
import importlib

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


def test_one_rank_gather_of_frames_and_bands(gsx_fresh, pkg, oracle, synth):
    g = gsx_fresh
    hip = _hip()
    mg = importlib.import_module("gaussian_splatting_amd.multigpu")
    cam = importlib.import_module("gaussian_splatting_amd.camera")
    W, H = 320, 200
    fb_bytes = W * H * 4
    splats = oracle.raw_to_splat(synth.clustered_cloud(40_000, seed=11))
    g.upload(splats)
    view, proj, pos = cam.camera_matrices(cam.orbit_world(5.0, 0.4, 0.2), 45.0, W / H)
    u = pkg.make_uniforms(proj, view)
    g.sort(pos, download=False)
    want = g.render(u, W, H).copy()

    g.comm_init(None, 0, 1)                       # one rank: no RCCL needed, the gather is a device copy
    recv = C.c_void_p()
    assert hip.hipMalloc(C.byref(recv), 3 * fb_bytes) == 0
    assert hip.hipMemset(recv, 0xAB, 3 * fb_bytes) == 0
    # (1) a frame that is still queued (no download, no sync) is gathered at an offset: the gather is ordered behind it
    g.sort(pos, download=False)
    g.render(u, W, H, download=False)
    fb = g.framebuffer_dev()[0]
    g.gather(fb, fb_bytes, recv.value, [fb_bytes], [fb_bytes], 0)
    g.sync()
    got = np.empty(3 * fb_bytes, np.uint8)
    assert hip.hipMemcpy(got.ctypes.data, recv.value, 3 * fb_bytes, 2) == 0
    assert np.array_equal(got[fb_bytes:2 * fb_bytes], want.reshape(-1))
    assert np.all(got[:fb_bytes] == 0xAB) and np.all(got[2 * fb_bytes:] == 0xAB)
    # (2) bands gathered in place (send == recv + offset): nothing moves, nothing is damaged
    bands = mg.shard_bands(H, 1)
    g.sort(pos, download=False)
    g.render(u, W, H, download=False)
    mg.gather_bands_gsx(g, fb, bands, W, 0)
    g.sync()
    own = np.empty(fb_bytes, np.uint8)
    assert hip.hipMemcpy(own.ctypes.data, fb, fb_bytes, 2) == 0
    assert np.array_equal(own, want.reshape(-1))
    # (3) view batches through GsxFrameGather: frames rendered straight into batch slots
    keep = []

    def alloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), nbytes) == 0
        keep.append(p)
        return p.value, p

    fg = mg.GsxFrameGather(g, 0, 1, fb_bytes, 2, alloc)
    poses = [cam.camera_matrices(cam.orbit_world(5.0, 0.3 * i, 0.1), 45.0, W / H) for i in range(3)]
    refs = []
    for i, (v, p, c) in enumerate(poses):
        g.sort(c, download=False)
        g.set_framebuffer_dev(fg.slot(i), fb_bytes)
        if i:
            fg.frame_done(i - 1)
        g.render(pkg.make_uniforms(p, v), W, H, download=False)
    g.set_framebuffer_dev(None)
    fg.frame_done(2)
    fg.finish()
    g.sync()
    assert fg.gathers == 2
    for i, (v, p, c) in enumerate(poses):
        g.sort(c, download=False)
        refs.append(g.render(pkg.make_uniforms(p, v), W, H).reshape(-1).copy())
    for i, (b, s) in enumerate(((0, 0), (0, 1), (1, 0))):
        out = np.empty(fb_bytes, np.uint8)
        assert hip.hipMemcpy(out.ctypes.data, fg.frame_on_rank0(b, 0, s), fb_bytes, 2) == 0
        assert np.array_equal(out, refs[i]), f"frame {i}"
    # (4) argument checking
    with pytest.raises(pkg.GsxError) as e:
        g.gather(fb, fb_bytes, recv.value, [0], [fb_bytes - 4], 0)        # root's size differs from what it sends
    assert e.value.code == pkg.GSX_E_INVALID
    with pytest.raises(pkg.GsxError):
        g.gather(fb, fb_bytes, None, None, None, 0)                        # root without a receive buffer
    with pytest.raises(pkg.GsxError):
        g.gather(fb, fb_bytes, recv.value, [0], [fb_bytes], 1)             # root outside the communicator
    with pytest.raises(pkg.GsxError):
        g.comm_init(None, 0, 2)                                            # more than one rank needs an id
    g.comm_destroy()
    for p in keep:
        assert hip.hipFree(p) == 0
    assert hip.hipFree(recv) == 0


def test_exchange_runs_beside_frames_in_flight(gsx_fresh, pkg, oracle, synth):
    """The exchange has a stream of its own (include/gsx.h, gsx_gather): (a) a frame rendered into memory that an exchange
    still ships waits for it on the device, so back-to-back render / gather / render into ONE buffer, never synchronized
    by the caller, delivers every frame intact; (b) exchanges posted two frames late (GsxFrameGather's steady path: the
    host never waits) ship the right frames while later ones are in flight."""
    g = gsx_fresh
    hip = _hip()
    mg = importlib.import_module("gaussian_splatting_amd.multigpu")
    cam = importlib.import_module("gaussian_splatting_amd.camera")
    W, H = 640, 400
    fb_bytes = W * H * 4
    splats = oracle.raw_to_splat(synth.clustered_cloud(200_000, seed=5))
    g.upload(splats)
    poses = [cam.camera_matrices(cam.orbit_world(5.0, 0.05 * i, 0.1), 45.0, W / H) for i in range(9)]
    unis = [pkg.make_uniforms(p, v) for v, p, _ in poses]
    refs = []
    for i, (v, p, c) in enumerate(poses):
        g.sort(c, download=False)
        refs.append(g.render(unis[i], W, H).reshape(-1).copy())
    g.comm_init(None, 0, 1)
    keep = []

    def alloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), nbytes) == 0
        assert hip.hipMemset(p, 0xCD, nbytes) == 0
        keep.append(p)
        return p.value, p

    def read(ptr, nbytes=fb_bytes):
        out = np.empty(nbytes, np.uint8)
        assert hip.hipMemcpy(out.ctypes.data, C.c_void_p(ptr), nbytes, 2) == 0
        return out

    # (a) one target, three frames, three exchanges, no synchronization in between
    x, _ = alloc(fb_bytes)
    r, _ = alloc(3 * fb_bytes)
    g.set_framebuffer_dev(x, fb_bytes)
    for i in range(3):
        g.sort(poses[i][2], download=False)
        g.render(unis[i], W, H, download=False)
        g.gather(x, fb_bytes, r, [i * fb_bytes], [fb_bytes], 0)
    g.set_framebuffer_dev(None)
    g.sync()
    for i in range(3):
        assert np.array_equal(read(r + i * fb_bytes), refs[i]), f"frame {i} was overwritten before its exchange had read it"
    assert np.array_equal(read(x), refs[2])

    # (b) batches of 2, exchanges posted 2 frames late, 9 frames -> 5 exchanges
    fg = mg.GsxFrameGather(g, 0, 1, fb_bytes, 2, alloc)
    assert fg.lag == 2
    for i in range(9):
        g.sort(poses[i][2], download=False)
        g.set_framebuffer_dev(fg.slot(i), fb_bytes)
        if i:
            fg.frame_done(i - 1)
        g.render(unis[i], W, H, download=False)
        assert fg.gathers == max(0, (i - 2) // 2)     # batch k leaves when frame 2k + 3 has been queued
    g.set_framebuffer_dev(None)
    fg.frame_done(8)
    fg.finish()
    g.sync()
    assert fg.gathers == 5
    assert np.array_equal(read(fg.frame_on_rank0(1, 0, 0)), refs[6]) and np.array_equal(read(fg.frame_on_rank0(1, 0, 1)), refs[7])
    assert np.array_equal(read(fg.frame_on_rank0(0, 0, 0)), refs[8])
    assert np.array_equal(read(fg.frame_on_rank0(0, 0, 1)), refs[5])      # the stale slot of the partial last batch
    # a batch of one frame: the lag shrinks to 1 (the buffer is refilled two frames later)
    fg1 = mg.GsxFrameGather(g, 0, 1, fb_bytes, 1, alloc)
    assert fg1.lag == 1
    for i in range(5):
        g.sort(poses[i][2], download=False)
        g.set_framebuffer_dev(fg1.slot(i), fb_bytes)
        if i:
            fg1.frame_done(i - 1)
        g.render(unis[i], W, H, download=False)
    g.set_framebuffer_dev(None)
    fg1.frame_done(4)
    fg1.finish()
    g.sync()
    assert fg1.gathers == 5
    assert np.array_equal(read(fg1.frame_on_rank0(0, 0, 0)), refs[4]) and np.array_equal(read(fg1.frame_on_rank0(1, 0, 0)), refs[3])
    g.comm_destroy()
    for p in keep:
        assert hip.hipFree(p) == 0
