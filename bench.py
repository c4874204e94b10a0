#!/usr/bin/env python3
"""bench.py — headline benchmark of the gsx hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

Workload (BASELINE.json configs[1]): 6M-splat synthetic clustered scene, one
1920x1080 view per step.  A *step* is one complete frame of the hot path on
data already resident in HBM: depth keys -> stable radix sort -> projection ->
near-to-far iterations of (occlusion-aware tile binning -> pair sort -> tile raster)
into a device RGBA8 framebuffer (no readback).
At N > 1 the path shards by camera view over a replicated splat set (weak
scaling: a step is a batch of N views, one per rank, every rank at the pose of the N = 1 run so that the per-GPU work is
fixed; --distinct-views: rank r renders pose r of the pose set) and the finished frames are
gathered to rank 0 with RCCL inside the timed region: every rank renders into the
slots of a batch buffer and each full batch of 8 frames goes out in one asynchronous
gather while the next batch is rendered (multigpu.BatchedFrameGather).

Prints ONE JSON line on rank 0.  `value` = splats sorted (and drawn) per second
over whole frames, aggregated over ranks; `fps_1080p` and the sort-only rate are
reported next to it.  `roofline` describes the radix scatter kernel (k_radix_scatter<u32>): inside the
timed region ONE of its launches per frame is bracketed by HIP events on the library's stream (the pass
rotates frame by frame; bracketing every stage costs ~65 us per frame and is done on a few extra frames
AFTER the timed region for `stage_ms`).  `cpu_baseline` times the CPU port of the reference's sortSplats
on rank 0.
"""
import argparse
import importlib
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SCATTER_BYTES_PER_SPLAT = 16    # one k_radix_scatter launch: read key+index (8 B), write key+index (8 B)
KEYGEN_BYTES_PER_SPLAT = 20     # float4 position read + u32 key write (pass-0 digit count fused, no extra bytes)
COUNT_BYTES_PER_SPLAT = 4       # k_radix_hist of passes >= 1: key read


def sort_bytes_per_splat(passes):
    """Algorithmic HBM bytes per splat of keys + sort: SURVEY 8(d) counts 92 B for
    a 4-pass onesweep; this design runs `passes` scatter passes plus a 4-B count read for every pass but
    the first."""
    return KEYGEN_BYTES_PER_SPLAT + passes * SCATTER_BYTES_PER_SPLAT + max(passes - 1, 0) * COUNT_BYTES_PER_SPLAT


def load_pkg():
    name = "gaussian_splatting_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg_dir = os.path.join(ROOT, "gaussian-splatting_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def orbit_views(pkg, n_views, width, height):
    """8 azimuths x 8 elevations style pose set at distance 5 (SURVEY 8d cfg 4), any count."""
    cam = importlib.import_module("gaussian_splatting_amd.camera")
    views = []
    for v in range(n_views):
        az = 2.0 * np.pi * (v % 8) / 8.0 + 0.05 * (v // 64)
        el = np.deg2rad(-35.0 + 10.0 * ((v // 8) % 8))
        world = cam.orbit_world(5.0, az, el)
        view, proj, pos = cam.camera_matrices(world, 45.0, width / height)
        views.append((view, proj, pos))
    return views


class DevArray:
    """Zero-copy torch view of a raw device pointer (framebuffer) via __cuda_array_interface__."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": (nbytes,), "typestr": "|u1", "version": 2}


def cpu_baseline(splats, views, n_frames):
    """CPU port of SplatMesh::sortSplats (oracle, test infrastructure) on this host, 1 core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    O = orc.Oracle()
    idx = np.arange(splats.shape[0], dtype=np.uint32)   # persistent across frames, like SplatMesh::indices
    secs = 0.0
    for f in range(n_frames):
        secs += O.sort_reference_port(splats, views[f % len(views)][2], idx)
    return splats.shape[0] * n_frames / secs, secs


def cpu_baseline_parallel(splats, views, n_frames, threads):
    """Multi-threaded stand-in for the reference's PARALLEL build (tbb::task_arena(8) + std::sort(par), SplatMesh.h:97-103):
    libstdc++ parallel-mode sort on `threads` OpenMP threads (oracle/orc_parsort.cpp; the image has no TBB headers)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    O = orc.Oracle()
    idx = np.arange(splats.shape[0], dtype=np.uint32)
    secs = 0.0
    for f in range(n_frames):
        secs += O.sort_reference_port_parallel(splats, views[f % len(views)][2], idx, threads)
    return splats.shape[0] * n_frames / secs, secs


def _dbg(msg):
    if os.environ.get("GSX_BENCH_DEBUG"):
        print(f"[bench {time.perf_counter():.3f}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--splats", type=int, default=6_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="clustered")
    ap.add_argument("--cpu-frames", type=int, default=20, help="frames of the CPU baseline (0 = skip)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--distinct-views", action="store_true", help="N > 1: rank r renders pose r instead of the N = 1 pose")
    ap.add_argument("--gather-batch", type=int, default=8, help="frames per RCCL gather at N > 1")
    ap.add_argument("--opt", action="append", default=[], help="name=value for gsx_set_option (A/B runs; not used by the driver)")
    a = ap.parse_args()

    # stdout carries exactly ONE line, the JSON: libraries that print banners to stdout (RCCL prints its version when the
    # communicator is created) are sent to stderr by swapping the descriptors before anything is loaded
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1 or os.environ.get("GSX_BENCH_FORCE_DIST"):   # (the variable: one-rank rehearsal of the N > 1 code path)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    _dbg(f"process group ready: rank {rank}/{world}")

    pkg = load_pkg()
    synth = importlib.import_module("gaussian_splatting_amd.synth")
    scene = importlib.import_module("gaussian_splatting_amd.scene")
    raw = synth.SCENES[a.scene](a.splats)
    splats = scene.raw_to_splats(pkg, raw, center=False)      # host half of ResourceManager (libgsx, C++)
    W, H = a.width, a.height
    views = orbit_views(pkg, 64, W, H)
    # view-batch sharding (weak scaling): a step of the job is a batch of N views, one per GPU (independent units, no
    # data-path collective until the finished frames are gathered)
    mg = importlib.import_module("gaussian_splatting_amd.multigpu")
    # Poses differ a lot in cost on this scene (0.41 .. 0.74 ms over the first 8, DESIGN.md section 5: tile pairs, and a camera
    # inside the scene box needs a 4th radix pass), which would show up as a scaling loss that has nothing to do with N.
    # Weak scaling wants the per-GPU work fixed, so by default every rank renders the pose of the N = 1 run into its own
    # frame buffers; --distinct-views gives rank r pose r of the set instead.
    my_views = [views[mg.shard_views(len(views), rank, max(world, 1))[0] if a.distinct_views else 0]]
    g = pkg.Gsx(local_rank)
    g.upload(splats)
    for o in a.opt:
        k, v = o.split("=")
        g.set_option(k, int(v))
    g.set_profiling(2)          # sampled: one k_radix_scatter launch per frame carries HIP events
    uniforms = [pkg.make_uniforms(v[1], v[0]) for v in my_views]

    fb_bytes = W * H * 4
    # N > 1: every rank renders straight into the slots of a batch buffer; a full batch (8 frames, SURVEY 8e cfg 4)
    # goes to rank 0 in one asynchronous RCCL gather while the next batch is rendered (two buffers alternate).
    bg = None
    if dist is not None and not a.no_gather:
        ext = torch.cuda.ExternalStream(g.stream(), device=torch.device("cuda", local_rank))
        bg = mg.BatchedFrameGather(dist, rank, world, fb_bytes, a.gather_batch, torch.device("cuda", local_rank), stream=ext)
    frame_no = 0

    def step(i):
        nonlocal frame_no
        v = i % len(my_views)
        g.sort(my_views[v][2], download=False)          # queued behind the running frame
        if bg is not None:
            slot = bg.slot(frame_no)
            g.set_framebuffer_dev(slot.data_ptr(), fb_bytes)   # completes the previous frame into ITS slot first
            if frame_no:
                bg.frame_done(frame_no - 1)             # closes a batch every gather_batch frames: async gather
        g.render(uniforms[v], W, H, download=False)
        frame_no += 1

    def drain():
        """Complete the last frame and send what is left of the current batch; waits for every gather."""
        if bg is not None and frame_no:
            g.set_framebuffer_dev(None)
            bg.frame_done(frame_no - 1)
            bg.finish()

    def fence():
        if dist is not None:
            dist.barrier()
        g.sync()
        torch.cuda.synchronize()

    _dbg("scene uploaded")
    for i in range(a.warmup):
        step(i)
    fence()
    _dbg("warm-up done")
    g.scatter_samples(reset=True)
    # Frames are queued back to back (the next frame's sort is queued behind the running frame; gsx_render waits for
    # the previous frame only to check whether it needs more iterations).  Nothing is read back inside the loop: the
    # sampled scatter launches sit in the library's event ring and are collected after the closing fence.
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
        if i % 16 == 15:
            _dbg(f"queued step {i}")
    _dbg("timed steps queued")
    drain()
    _dbg("drained")
    fence()
    elapsed = time.perf_counter() - t0
    _dbg("timed region closed")
    bg = None                                            # the extra profiling frames below are rank 0's alone: no collective
    per_pass, pass_n = g.scatter_samples(reset=True)     # mean launch duration of each radix pass over the timed region
    t = g.timings()
    pairs = t["tile_pairs"] * a.steps; iters = t["render_iters"] * a.steps; passes = t["sort_passes"] * a.steps
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # stage breakdown on a few extra frames with every stage bracketed (outside the timed region)
    acc = {}
    n_prof = 6
    if rank == 0:
        g.set_profiling(1)
        for i in range(n_prof + 1):
            step(i)
            t = g.timings()
            if i:                                  # first frame after switching levels is a warm-up
                for k, val in t.items():
                    acc[k] = acc.get(k, 0.0) + val

    if rank == 0:
        K = a.steps
        n = a.splats
        passes_per_frame = passes / K
        launch_ms = sum(per_pass.values()) / max(len(per_pass), 1)       # average k_radix_scatter<u32> launch
        achieved = SCATTER_BYTES_PER_SPLAT * n / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        keysort_ms = (acc["keygen_ms"] + acc["sort_ms"]) / n_prof
        sort_bytes = sort_bytes_per_splat(int(round(passes_per_frame)))
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r1_scatter_traffic.json")     # PMC passes (tools/pmc_traffic.sh), committed
        if os.path.exists(tf):
            rec = json.load(open(tf))
            if rec.get("splats") == n:
                traffic = rec.get("hbm_bytes_per_launch")
        out = {
            "metric": "splats_sorted_per_s (whole 1080p frame: keys+sort+project+bin+raster)",
            "value": n * K * world / elapsed,
            "unit": "splats/s",
            "n_gpus": world,
            "steps": K,
            "warmup": a.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 keys / u32 radix",
            "data": "synthetic",
            "config": {"workload": f"cfg2: {n}-splat clustered synthetic scene, single view {W}x{H} per step",
                       "splats": n, "width": W, "height": H, "views_per_step_per_gpu": 1,
                       "poses": "rank r renders pose r" if a.distinct_views else "every rank renders the N=1 pose (fixed per-GPU work)",
                       "sharding": "camera views over replicated splats" + ("" if a.no_gather or world == 1 else f" + RCCL framebuffer gather to rank 0 in batches of {a.gather_batch} frames")},
            "fps_1080p": K * world / elapsed,
            "sort_only_splats_per_s": n / (keysort_ms * 1e-3),
            "sort_only_ms": keysort_ms,
            "sort_only_bytes_per_splat": sort_bytes,
            "sort_only_hbm_frac": sort_bytes * n / (keysort_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "sort_only_hbm_frac_at_92B": 92 * n / (keysort_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "stage_ms": {k: acc[k] / n_prof for k in ("select_ms", "keygen_ms", "sort_ms", "scatter_ms", "project_ms", "bin_ms", "raster_ms", "total_ms")},
            "stage_ms_note": f"{n_prof} extra frames after the timed region with every stage bracketed by HIP events (adds ~65 us per frame)",
            "tile_pairs": int(pairs / K),
            "render_iters": iters / K,
            "roofline": {"kernel": "k_radix_scatter<u32> (one stable radix scatter pass of the depth sort)", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": launch_ms, "per_pass_launch_ms": [per_pass[p] for p in sorted(per_pass)],
                         "launches_per_step": passes_per_frame, "launches_timed": int(sum(pass_n.values())),
                         "algorithmic_bytes_per_launch": SCATTER_BYTES_PER_SPLAT * n},
        }
        if a.cpu_frames > 0:
            v, secs = cpu_baseline(splats, views, a.cpu_frames)
            out["cpu_baseline"] = {"value": v, "unit": "splats/s", "cores": 1, "kind": "port",
                                   "sample": f"{a.cpu_frames} consecutive frames of the full {n}-splat sortSplats "
                                             f"(distances alloc + key loop + std::sort, {secs:.1f} s of CPU; sort only, the "
                                             "reference has no CPU raster)"}
            try:
                threads = 8                                      # the reference's fixed arena size (SplatMesh.h:98)
                v, secs = cpu_baseline_parallel(splats, views, a.cpu_frames, threads)
                out["cpu_baseline_parallel"] = {"value": v, "unit": "splats/s", "cores": threads, "kind": "port",
                                                "sample": f"{a.cpu_frames} consecutive frames of the full {n}-splat sortSplats, "
                                                          f"{secs:.1f} s wall; libstdc++ parallel-mode sort (OpenMP) with the "
                                                          "reference's comparator on the reference's 8 threads -- NOT TBB: "
                                                          "the image has no TBB headers, so the PARALLEL build cannot be compiled"}
            except OSError as e:                                 # liborc_par.so / libgomp missing: say so, keep the line
                out["cpu_baseline_parallel"] = {"value": None, "note": f"unavailable: {e}"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    _dbg("closing")
    g.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
