"""CPU, world size 2, gloo: the N > 1 path of bench.py — view sharding, tile-band sharding and the
framebuffer gather to rank 0 — with the oracle standing in for the GPU renderer."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

WORKER = r'''
import importlib, importlib.util, os, sys
import numpy as np, torch, torch.distributed as dist
ROOT = sys.argv[1]
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
import orc
pkg = load_pkg()
mg = importlib.import_module("gaussian_splatting_amd.multigpu")
synth = importlib.import_module("gaussian_splatting_amd.synth")
cam = importlib.import_module("gaussian_splatting_amd.camera")
scene = importlib.import_module("gaussian_splatting_amd.scene")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
O = orc.Oracle()
W, H = 96, 72
splats = scene.raw_to_splats(pkg, synth.uniform_cloud(600, seed=5, half_extent=1.5))
views = []
for v in range(4):
    world_m = cam.orbit_world(5.0, 0.5 * v, 0.1 * v)
    views.append(cam.camera_matrices(world_m, 45.0, W / H))
def render(v, y0=0, y1=H):
    view, proj, pos = views[v]
    perm = O.sort_contract(splats, pos)
    return O.render(splats, perm, orc.make_uniforms(proj, view), W, H, y0, y1)
# --- view sharding + gather ---
mine = mg.shard_views(len(views), rank, world)
ok = True
for step in range(len(views) // world):
    fb = torch.from_numpy(render(mine[step]).reshape(-1).copy())
    got = mg.gather_to_rank0(dist, fb, rank, world)
    if rank == 0:
        for r in range(world):
            ref = render(mg.shard_views(len(views), r, world)[step]).reshape(-1)
            ok &= bool(np.array_equal(got[r].numpy(), ref))
# --- band sharding + gather + compose ---
bands = mg.shard_bands(H, world)
y0, y1 = bands[rank]
fb = torch.from_numpy(render(1, y0, y1).reshape(-1).copy())
got = mg.gather_to_rank0(dist, fb, rank, world)
if rank == 0:
    whole = mg.compose_bands(got, bands, W).numpy()
    ok &= bool(np.array_equal(whole, render(1).reshape(-1)))
# --- batched asynchronous gather (bench.py at N > 1): 7 frames per rank in batches of 3, double-buffered ---
FB, B, NF = 64, 3, 7
def frame(r, i):
    return torch.full((FB,), (17 * r + 3 * i + 1) % 251, dtype=torch.uint8)
bg = mg.BatchedFrameGather(dist, rank, world, FB, B, "cpu")
for i in range(NF):
    bg.slot(i).copy_(frame(rank, i))
    bg.frame_done(i)
bg.finish()
ok &= bg.gathers == 3 and all(w is None for w in bg.work)
if rank == 0:
    # frames 3..5 went through buffer 1, frame 6 (partial batch) through buffer 0 slot 0 on top of frames 0..2
    for r in range(world):
        got1 = [int(t[0]) for t in bg.frames_on_rank0(1)[r]]
        ok &= got1 == [int(frame(r, i)[0]) for i in (3, 4, 5)]
        got0 = [int(t[0]) for t in bg.frames_on_rank0(0)[r]]
        ok &= got0 == [int(frame(r, 6)[0]), int(frame(r, 1)[0]), int(frame(r, 2)[0])]
# --- the gsx_gather forms (bench.py on GPUs): same argument contract as include/gsx.h gsx_gather, carried by gloo
#     send / recv over a byte heap whose offsets stand in for device pointers ---
class HeapGsx:
    BASE = 4096
    def __init__(self, nbytes):
        self.heap = torch.zeros(nbytes, dtype=torch.uint8)
        self.top = 0
    def alloc(self, nbytes):
        p = self.BASE + self.top
        self.top += (nbytes + 255) // 256 * 256
        assert self.top <= self.heap.numel()
        return p, None
    def mem(self, p, n):
        return self.heap[p - self.BASE:p - self.BASE + n]
    def gather(self, send_ptr, send_bytes, recv_ptr=None, recv_offsets=None, recv_bytes=None, root=0):
        if rank == root:
            assert recv_bytes[root] == send_bytes
            if recv_ptr + recv_offsets[root] != send_ptr and send_bytes:
                self.mem(recv_ptr + recv_offsets[root], send_bytes).copy_(self.mem(send_ptr, send_bytes).clone())
            for r in range(world):
                if r != root and recv_bytes[r]:
                    dist.recv(self.mem(recv_ptr + recv_offsets[r], recv_bytes[r]), src=r)
        elif send_bytes:
            dist.send(self.mem(send_ptr, send_bytes).contiguous(), dst=root)
hg = HeapGsx(1 << 20)
# bands, gathered in place into rank 0's frame
fb_ptr, _ = hg.alloc(W * H * 4)
hg.mem(fb_ptr, W * H * 4).copy_(torch.from_numpy(render(2, y0, y1).reshape(-1).copy()))
mg.gather_bands_gsx(hg, fb_ptr, bands, W, rank)
if rank == 0:
    ok &= bool(np.array_equal(hg.mem(fb_ptr, W * H * 4).numpy(), render(2).reshape(-1)))
off, size = mg.band_gather_plan(bands, W)
ok &= off[0] == 0 and sum(size) == W * H * 4 and all(o + s == o2 for o, s, o2 in zip(off, size, off[1:]))
# view batches: 7 frames per rank in batches of 3
fgx = mg.GsxFrameGather(hg, rank, world, FB, B, hg.alloc)
for i in range(NF):
    hg.mem(fgx.slot(i), FB).copy_(frame(rank, i))
    fgx.frame_done(i)
fgx.finish()
ok &= fgx.gathers == 3
if rank == 0:
    for r in range(world):
        ok &= [int(hg.mem(fgx.frame_on_rank0(1, r, s), FB)[0]) for s in range(B)] == [int(frame(r, i)[0]) for i in (3, 4, 5)]
        ok &= [int(hg.mem(fgx.frame_on_rank0(0, r, s), FB)[0]) for s in range(B)] == [int(frame(r, i)[0]) for i in (6, 1, 2)]
    print("RESULT", "OK" if ok else "MISMATCH")
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_helpers(pkg):
    import importlib
    mg = importlib.import_module("gaussian_splatting_amd.multigpu")
    assert mg.shard_views(64, 3, 8) == list(range(3, 64, 8))
    assert sorted(sum((mg.shard_views(10, r, 4) for r in range(4)), [])) == list(range(10))
    for h, w in ((2160, 8), (1080, 8), (72, 2), (16, 4), (100, 3)):
        bands = mg.shard_bands(h, w)
        assert bands[0][0] == 0 and bands[-1][1] == h
        for (a0, a1), (b0, b1) in zip(bands, bands[1:]):
            assert a1 == b0 and a0 <= a1
        assert all(y0 % 16 == 0 for y0, _ in bands)
    assert mg.shard_bands(2160, 8)[0] == (0, 272)        # 135 tile rows over 8 ranks: 17,17,17,17,17,17,17,16


def test_frame_gather_posts_exchanges_late_and_in_order(pkg):
    """GsxFrameGather (the view-batch loop of bench.py at N > 1), on a recording stand-in for the library: the exchange of a
    batch is posted `lag` = min(2, batch) frames after the batch is complete - by then the library has finished its frames, so
    gsx_gather never waits on the host - always before the first frame that refills its buffer, in batch order, and finish()
    posts what is left, a partial batch included."""
    import importlib
    mg = importlib.import_module("gaussian_splatting_amd.multigpu")

    class Rec:
        def __init__(self):
            self.calls, self.base = [], 0x1000

        def gather(self, send, nbytes, recv, off, size, root):
            self.calls.append((send, nbytes, recv, None if off is None else list(off), root))

        def alloc(self, nbytes):
            p = self.base
            self.base += (nbytes + 255) // 256 * 256
            return p, None

    FB = 64
    for batch, frames in ((1, 5), (2, 9), (3, 7), (8, 20)):
        for rank, world in ((0, 1), (0, 4), (3, 4)):
            g = Rec()
            fg = mg.GsxFrameGather(g, rank, world, FB, batch, g.alloc)
            lag = min(2, batch)
            assert fg.lag == lag
            posted_at = {}                                  # batch index -> frame after which its exchange was posted
            for i in range(frames):
                slot = fg.slot(i)
                b = (i // batch) % 2
                assert slot == fg.batches[b] + (i % batch) * FB
                # the buffer this frame is rendered into is not waiting for an exchange of an earlier batch
                first_of_batch = (i // batch) * batch
                if i == first_of_batch and i >= 2 * batch:
                    assert (i // batch) - 2 in posted_at, (batch, i)
                before = len(g.calls)
                fg.frame_done(i)
                for k in range(before, len(g.calls)):
                    posted_at[k] = i
                    assert i == (k + 1) * batch - 1 + lag     # exactly `lag` frames behind the batch's last frame
            fg.finish()
            n_batches = (frames + batch - 1) // batch
            assert fg.gathers == len(g.calls) == n_batches
            for k, (send, nbytes, recv, off, root) in enumerate(g.calls):
                assert send == fg.batches[k % 2] and nbytes == batch * FB and root == 0
                if rank == 0:
                    assert recv == fg.recv[k % 2] and off == [r * batch * FB for r in range(world)]
                else:
                    assert recv is None and off is None
            fg.finish()                                      # idempotent
            assert len(g.calls) == n_batches
            # a rank that renders fewer frames than the longest one posts up to the collective batch count (same sizes,
            # alternating buffers), so that every rank posts the same number of exchanges
            fg.finish(total_batches=n_batches + 2)
            assert len(g.calls) == n_batches + 2
            assert [c[0] for c in g.calls[-2:]] == [fg.batches[n_batches % 2], fg.batches[(n_batches + 1) % 2]]


def test_two_rank_gloo_view_and_band_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    port = 29500 + (os.getpid() % 2000)
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert "RESULT OK" in outs[0][0], outs[0][0] + outs[0][1][-1000:]
