"""Sharding of the path across GPUs (one process + one gsx context per GPU, splats replicated).

Two natural shardings, no data-path collective until the frames are finished:
  * camera views   : rank r renders views r, r + R, r + 2R, ...
  * tile-row bands : rank r renders rows [y0, y1) of one frame, band edges on tile rows
The single exchange is the gather of RGBA8 pixels to rank 0.  On GPUs it goes through the library's own entry point
(gsx_gather: grouped RCCL send / recv on the context's exchange stream, GsxFrameGather / gather_bands_gsx below); the
torch.distributed forms (gather_to_rank0, BatchedFrameGather) are the harness for the gloo CPU tests and the fallback.
"""
TILE_H = 16


def shard_views(n_views, rank, world):
    """Indices of the views rank `rank` renders (round-robin keeps per-rank load equal)."""
    return list(range(rank, n_views, world))


def shard_bands(height, world, tile_h=TILE_H):
    """[(y0, y1)] per rank: contiguous bands aligned to tile rows, covering [0, height) exactly.
    Ranks beyond the number of tile rows get an empty band (y0 == y1)."""
    rows = (height + tile_h - 1) // tile_h
    bands, start = [], 0
    for r in range(world):
        n = rows // world + (1 if r < rows % world else 0)
        y0, y1 = min(start * tile_h, height), min((start + n) * tile_h, height)
        bands.append((y0, y1))
        start += n
    return bands


def gather_to_rank0(dist, tensor, rank, world, bufs=None, group=None):
    """Gather equal-sized uint8 tensors to rank 0.  Returns the list on rank 0, None elsewhere."""
    import torch
    if rank == 0:
        if bufs is None:
            bufs = [torch.empty_like(tensor) for _ in range(world)]
        dist.gather(tensor, bufs, dst=0, group=group)
        return bufs
    dist.gather(tensor, None, dst=0, group=group)
    return None


def compose_bands(frames, bands, width):
    """Rank-0 side of band sharding: frames[r] is rank r's full-size RGBA8 buffer with its band filled."""
    out = frames[0].clone()
    row = width * 4
    for r, (y0, y1) in enumerate(bands):
        if r and y1 > y0:
            out[y0 * row:y1 * row] = frames[r][y0 * row:y1 * row]
    return out


class BatchedFrameGather:
    """View-batch sharding, the exchange step (cfg 4): every rank renders its views straight into the
    slots of a batch buffer and, once `batch_frames` frames are complete, the whole batch goes to rank 0 in ONE
    gather (direct sends, each peer on its own xGMI link).  Two batch buffers alternate, and the gather is started
    asynchronously, so the next batch is rendered while RCCL moves the previous one.

    `stream` (optional): a torch.cuda stream the renderer queues its work on (torch.cuda.ExternalStream of the
    library's HIP stream).  The gather is then ordered behind the frames already queued on it, and re-using a
    batch buffer makes that stream (not the host) wait for the gather that still reads it.  Without a stream
    (gloo / CPU tensors) the waits are host-side.

    Per frame i (0, 1, 2 ... of this rank):  t = slot(i) -> render into t -> frame_done(i) once frame i has been
    queued completely.  finish() sends the last, possibly partial, batch and waits for everything."""

    def __init__(self, dist, rank, world, frame_bytes, batch_frames, device, stream=None, group=None):
        import torch
        self.dist, self.rank, self.world, self.group = dist, rank, world, group
        self.frame_bytes, self.batch_frames, self.stream = frame_bytes, batch_frames, stream
        self.batches = [torch.zeros(batch_frames * frame_bytes, dtype=torch.uint8, device=device) for _ in range(2)]
        self.recv = [[torch.zeros_like(b) for _ in range(world)] if rank == 0 else None for b in self.batches]
        self.work = [None, None]
        self.sent = [False, False]          # batch holds frames that have not been gathered yet
        self.gathers = 0

    def _ctx(self):
        import contextlib
        import torch
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _wait(self, b):
        if self.work[b] is not None:
            with self._ctx():
                self.work[b].wait()          # nccl: the current stream waits; gloo: the host waits
            self.work[b] = None

    def _start(self, b):
        with self._ctx():
            self.work[b] = self.dist.gather(self.batches[b], self.recv[b], dst=0, async_op=True, group=self.group)
        self.sent[b] = False
        self.gathers += 1

    def slot(self, i):
        """Device memory frame i is rendered into (a view of the batch buffer)."""
        b, s = (i // self.batch_frames) % 2, i % self.batch_frames
        if s == 0:
            self._wait(b)                    # the gather started two batches ago still reads this buffer
        self.sent[b] = True
        return self.batches[b][s * self.frame_bytes:(s + 1) * self.frame_bytes]

    def frame_done(self, i):
        if i % self.batch_frames == self.batch_frames - 1:
            self._start((i // self.batch_frames) % 2)

    def finish(self):
        for b in range(2):
            if self.sent[b]:
                self._start(b)
        for b in range(2):
            self._wait(b)

    def frames_on_rank0(self, b):
        """rank 0: [rank][slot] views of the batch buffer b received last (None elsewhere)."""
        if self.rank != 0:
            return None
        fb = self.frame_bytes
        return [[t[s * fb:(s + 1) * fb] for s in range(self.batch_frames)] for t in self.recv[b]]


def band_gather_plan(bands, width):
    """(offsets, sizes) in bytes of every rank's band inside a width x height RGBA8 frame: the recv_offsets / recv_bytes
    of gsx_gather, with which the bands land in place in rank 0's frame."""
    row = width * 4
    return [y0 * row for y0, _ in bands], [(y1 - y0) * row for y0, y1 in bands]


def gather_bands_gsx(g, fb_ptr, bands, width, rank, root=0):
    """Tile-band sharding, the exchange step: every rank has rendered rows [y0, y1) of the frame into ITS OWN full-size
    framebuffer at fb_ptr (device pointer); the bands are gathered into the root's framebuffer in place (the root's own
    band is already there).  Runs on the context's exchange stream, behind the frame that rendered into fb_ptr and beside
    the frames queued after it; call it one frame late (after the NEXT frame has been queued) and the host never waits."""
    off, size = band_gather_plan(bands, width)
    g.gather(fb_ptr + off[rank], size[rank], fb_ptr if rank == root else None, off if rank == root else None,
             size if rank == root else None, root)


class GsxFrameGather:
    """BatchedFrameGather on gsx_gather: every rank renders into the slots of a batch buffer, a full batch goes to rank 0
    in one grouped RCCL exchange on the library's exchange stream, beside the frames of the next batch.  No host-side
    waits exist on the steady path: gsx_gather completes only the frames that render into the memory it ships, so the
    exchange of a batch is posted `lag` frames late (lag = 2, the number of frames the library keeps in flight; less when
    a batch is shorter) - by then the library has finished the batch's frames anyway - and a batch buffer is re-used by
    frames queued BEHIND the exchange that reads it (the library makes them wait for it on the device).  Two batch buffers
    alternate, so that on rank 0 the batch received last stays readable while the next one is filled.

    Per frame i (0, 1, 2 ... of this rank): render into slot(i), then frame_done(i) once it is queued (any time before
    frame i + 1 is rendered).  finish() posts what is left, a partial last batch included."""

    def __init__(self, g, rank, world, frame_bytes, batch_frames, alloc, lag=2):
        """alloc(nbytes) -> (device pointer, keep-alive object)."""
        self.g, self.rank, self.world = g, rank, world
        self.frame_bytes, self.batch_frames = frame_bytes, batch_frames
        self.lag = max(0, min(lag, batch_frames))     # (a batch's buffer is refilled 2 * batch_frames - lag ... frames later)
        nb = batch_frames * frame_bytes
        self.keep = []
        self.batches = []
        for _ in range(2):
            ptr, keep = alloc(nb)
            self.batches.append(ptr)
            self.keep.append(keep)
        self.recv = [None, None]
        if rank == 0:
            for b in range(2):
                ptr, keep = alloc(nb * world)
                self.recv[b] = ptr
                self.keep.append(keep)
        self.queued = 0               # frames queued so far
        self.next_batch = 0           # first batch whose exchange has not been posted
        self.gathers = 0

    def slot(self, i):
        b, s = (i // self.batch_frames) % 2, i % self.batch_frames
        return self.batches[b] + s * self.frame_bytes

    def _start(self, b):
        nb = self.batch_frames * self.frame_bytes
        # whole batch buffers are exchanged (a partial last batch carries stale slots: sizes must agree on all ranks)
        off = [r * nb for r in range(self.world)]
        size = [nb] * self.world
        root = self.rank == 0
        self.g.gather(self.batches[b], nb, self.recv[b] if root else None, off if root else None, size if root else None, 0)
        self.gathers += 1

    def frame_done(self, i):
        self.queued = max(self.queued, i + 1)
        # batch k is complete with frame (k + 1) * batch_frames - 1 and posted `lag` frames later
        while (self.next_batch + 1) * self.batch_frames - 1 + self.lag <= i:
            self._start(self.next_batch % 2)
            self.next_batch += 1

    def finish(self, total_batches=None):
        """Posts what is left.  EVERY rank must post the same number of exchanges (a grouped receive for a peer that never
        sends never completes): when the ranks render different numbers of frames - views that do not divide evenly - pass
        the collective batch count (the maximum over ranks of ceil(frames / batch_frames), e.g. an all-reduce MAX over the
        control group); a rank that is short posts its last buffers again (stale slots, same sizes)."""
        while self.next_batch * self.batch_frames < self.queued or (total_batches is not None and self.next_batch < total_batches):
            self._start(self.next_batch % 2)
            self.next_batch += 1

    def frame_on_rank0(self, b, r, s):
        """rank 0: device pointer of slot s of rank r's batch b as received last."""
        return self.recv[b] + (r * self.batch_frames + s) * self.frame_bytes
