#!/usr/bin/env python3
"""bench.py — headline benchmark of the gsx hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): 6M-splat synthetic clustered scene, one 1920x1080 view per step.  A *step* is
one complete frame of the hot path on data already resident in HBM: depth keys -> depth sort -> projection
-> near-to-far iterations of (occlusion-aware tile binning -> pair sort -> tile raster) into a device RGBA8
framebuffer (no readback).  EVERY STEP RENDERS A NEW CAMERA POSE (the 120-frame orbit of SURVEY 8d, 3 degrees apart,
starting at the default pose), so nothing a frame skips is known from an identical previous frame; the figure for a
camera at rest (same pose every step, where the library replays per-tile saturation ranks) is reported beside it as
`ms_per_step_static`.

N > 1, two shardings over a replicated splat set:
  --shard views (default; weak scaling): a step is a batch of N views, one per rank.  Every rank renders the pose of
      the N = 1 run so that the per-GPU work is fixed (--distinct-views: rank r renders pose r of the pose set).  The
      frames are rendered straight into the slots of a batch buffer and each full batch (--gather-batch, default 1 frame)
      goes to rank 0 in ONE gsx_gather (grouped RCCL send / recv on the library's exchange stream, beside the frames of the next batch)
      inside the timed region.
  --shard bands (strong scaling, BASELINE.json configs[4]): ONE 3840x2160 frame of the overlap-stress scene per step;
      rank r sorts the (replicated) scene -- with --band-cull only the splats whose quad can reach its band (gsx_select_band, a
      conservative cull BEFORE the sort) -- and renders tile band r, the bands are gathered in place into rank 0's frame, and after the timed region rank 0 checks that frame byte for byte against the frame it renders alone.
  --gather torch falls back to torch.distributed.gather; it is also used, and reported, if gsx_gather's start-up
  self-check fails.

Prints ONE JSON line on rank 0.  `value` = splats sorted (and drawn) per second over whole frames, aggregated over
ranks.  Beside the headline (a new pose every step) the line carries `ms_per_step_static` (one pose, history-driven
schedule and exact rank caps) and `ms_per_step_no_history` (one pose, every iteration pre-queued, no caps) at N = 1.
The depth sort of frame k+1 runs on a stream of its own beside the binning / rasterizing of frame k (include/gsx.h, option
"async_sort").  `roofline` describes the two HBM passes of the depth sort (k2_partition and k2_local; the radix scatter
kernel k_radix_scatter<u32> where the library runs its LSD form): inside the timed region every 5th (runs under 40 steps)
or steps/4-th (8..16) sort is a SAMPLED one -- two of three (three of four in the short runs) with every such launch of the sort
bracketed by its own pair of HIP events, the next with keys + sort bracketed as a whole -- on the stream it runs on, and that sort runs on the main stream, not beside the rasterizer,
so that a bracket measures the kernel itself rather than its share of a GPU it splits with another kernel.  A sampled sort
costs the two-frame pipeline about one frame time: `ms_per_step_unsampled` repeats the headline's loop at profiling level 0.
`roofline_raster`, the per-stage breakdown and the
rasterizer's shade / blend counts come from a few extra frames after the timed region with every stage bracketed.
`cpu_baseline` times the CPU port of the reference's sortSplats on rank 0.
"""
import argparse
import csv
import importlib
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_TFLOPS = 157.3        # f32 vector peak, packed or not (MI355X_MICROARCH.md, chip-level parameters)
LDS_PEAK_GBS = 256 * 256 * 2.4  # 256 B/clk/CU x 256 CUs x 2.4 GHz
SCATTER_BYTES_PASS0 = 12        # k_radix_scatter<u32, iota>: read key (4 B), write key + index (8 B)
SCATTER_BYTES_PASSN = 16        # passes >= 1: read key + index (8 B), write key + index (8 B)
KEYGEN_BYTES_PER_SPLAT = 20     # float4 position read + u32 key write (pass-0 digit count fused, no extra bytes)
COUNT_BYTES_PER_SPLAT = 4       # k_radix_hist of passes >= 1: key read
KEYGEN2_BYTES_PER_SPLAT = 16    # k2_keygen_hist: packed 12-byte position read + u32 key write (the 2-byte bucket counts ride along)
PARTITION_BYTES = 12            # k2_partition: read key (4 B; the index is generated), write (key, index) pair (8 B)
LOCAL_BYTES = 12                # k2_local: read pair (8 B), write index (4 B)
BLEND_FLOP = 14                 # per pixel-splat blend: uv (5), r2 + scale (4), alpha (1), weight (1), 3 colour FMAs + T (4), 1 exp beside
SHADE_LDS_BYTES = 48            # per (wave, splat) visit: centre/extents, uv matrix, colour records read from LDS


def sort_bytes_per_splat(passes, iota=True, form=1):
    """Algorithmic HBM bytes per splat of keys + sort."""
    if form == 2:       # keys + counts, partition, bucket sort (the count / prefix matrices add ~3 B per splat at 6 M: not counted)
        return KEYGEN2_BYTES_PER_SPLAT + PARTITION_BYTES + LOCAL_BYTES
    if passes <= 0:
        return KEYGEN_BYTES_PER_SPLAT
    first = SCATTER_BYTES_PASS0 if iota else SCATTER_BYTES_PASSN
    dead = 4 if passes < 4 else 0      # the last pass does not store its keys (unless a gated fourth pass sorts on them)
    return KEYGEN_BYTES_PER_SPLAT + first + (passes - 1) * (SCATTER_BYTES_PASSN + COUNT_BYTES_PER_SPLAT) - dead


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


def orbit_path(pkg, frames, width, height):
    """The 120-frame orbit of SURVEY 8d cfg 3: the orbit node rotated about +Y by 2 pi / frames per frame (Node::rotate)."""
    cam = importlib.import_module("gaussian_splatting_amd.camera")
    step = 2.0 * np.pi / frames
    out = []
    for s in range(frames):
        world = cam.orbit_pose(5.0, [((0.0, 1.0, 0.0), step)] * s)
        out.append(cam.camera_matrices(world, 45.0, width / height))
    return out


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


def kernel_rooflines_from_profile(n_splats):
    """HBM rates of the key / count / cull+projection / LOD kernels from the committed rocprofv3 --kernel-trace --stats summaries
    (tools/final_profile.sh runs this command once more with everything on ONE stream, so that the durations are the kernels'
    own; tools/lod_profile.sh profiles cfg 3).  Not measured in this run: every figure cites its file."""
    path = os.path.join(ROOT, "profiles", "r3_bench_serialized_kernel_stats.csv")
    if not os.path.exists(path):
        return None
    bytes_per_call = {   # algorithmic bytes per launch at n_splats splats
        "k2_keygen_hist": KEYGEN2_BYTES_PER_SPLAT * n_splats,     # 12 B position in, 4 B key out (the 2-byte bucket counts ride along)
        "k2_partition": PARTITION_BYTES * n_splats,
        "k2_local<18>": None,                                       # (the two bucket-sort instances share the pairs: priced together below)
        "k2_local<8>": None,
        "k2_scan": None,
    }
    out = {"source": "profiles/r3_bench_serialized_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python bench.py --cpu-frames 0 "
                     "--in-order --opt async_sort=0 --opt frames_in_flight=1`: one stream, one frame at a time; not measured in this run)"}
    local_us = 0.0
    for r in csv.DictReader(open(path)):
        for key, b in bytes_per_call.items():
            if key in r["Name"]:
                avg_us = float(r["AverageNs"]) / 1e3
                rec = {"avg_us": avg_us, "calls": int(r["Calls"])}
                if b:
                    rec.update(algorithmic_bytes=b, hbm_gbs=b / (avg_us * 1e-6) / 1e9, hbm_frac=b / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS)
                if key.startswith("k2_local"):
                    local_us += avg_us
                out[key] = rec
    if local_us > 0.0:
        b = LOCAL_BYTES * n_splats
        out["k2_local (both instances)"] = {"avg_us": local_us, "algorithmic_bytes": b, "hbm_gbs": b / (local_us * 1e-6) / 1e9,
                                            "hbm_frac": b / (local_us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    for name, fn in (("k_chunk_count", "r3_chunk_count_roofline.json"), ("lod_cut_cfg3", "r2_cfg3_lod_roofline.json"),
                     ("sort_pmc_traffic", "r3_sort_traffic.json")):
        p = os.path.join(ROOT, "profiles", fn)
        if os.path.exists(p):
            out[name] = dict(json.load(open(p)), source="profiles/" + fn)
    return out


def _dbg(msg):
    if os.environ.get("GSX_BENCH_DEBUG"):
        print(f"[bench {time.perf_counter():.3f}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--splats", type=int, default=6_000_000)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--scene", default=None)
    ap.add_argument("--shard", choices=("views", "bands"), default="views")
    ap.add_argument("--cpu-frames", type=int, default=20, help="frames of the CPU baseline (0 = skip)")
    ap.add_argument("--gather", choices=("gsx", "torch", "none"), default="gsx")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--distinct-views", action="store_true", help="N > 1, views: rank r renders pose r instead of the N = 1 pose")
    ap.add_argument("--gather-batch", type=int, default=0,
                    help="frames per gather at N > 1 (views); 0 = 1 for gsx_gather (the exchange stream overlaps every transfer with "
                         "the next frames anyway, and a short run then ends with one frame's transfer exposed instead of a batch's), "
                         "8 for the torch.distributed fallback (main-stream gathers)")
    ap.add_argument("--static-view", action="store_true", help="render the SAME pose every step (a camera at rest) instead of a new orbit pose per step")
    ap.add_argument("--band-cull", action="store_true",
                    help="bands: every rank first selects the splats whose quad can reach its band (gsx_select_band) and sorts only those; "
                         "measured per rank on cfg 5 it is time-neutral (0.22-0.28 ms per band either way, tools/configs_table.py cfg5: the splats of "
                         "that scene span 2-3 bands each and the sort runs beside the frame on its own stream), so the default sorts the replicated scene")
    ap.add_argument("--in-order", action="store_true", help="call order sort k, render k (no sort_ahead): the next sort is queued behind this frame's launches")
    ap.add_argument("--no-extras", action="store_true", help="skip the no-history / orbit / stage measurements after the timed region")
    ap.add_argument("--opt", action="append", default=[], help="name=value for gsx_set_option (A/B runs; not used by the driver)")
    a = ap.parse_args()
    if a.no_gather:
        a.gather = "none"
    bands_mode = a.shard == "bands"
    W = a.width or (3840 if bands_mode else 1920)
    H = a.height or (2160 if bands_mode else 1080)
    scene_name = a.scene or ("overlap" if bands_mode else "clustered")

    # stdout carries exactly ONE line, the JSON: libraries that print banners to stdout (RCCL prints its version when a
    # communicator is created) are sent to stderr by swapping the descriptors before anything is loaded
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("GSX_BENCH_FORCE_DIST"):
        # The library queues on four streams (main, sort, second frame slot, exchange) and RCCL brings streams of its own; with
        # HIP's default of 4 hardware queues per process they share queues and serialize (measured on one GPU with a torch NCCL
        # communicator in the process: +0.03 ms per frame at 4 queues, +0.00 at 8; no effect without one).  Read by HIP when
        # the runtime starts, so it is set before anything touches the GPU.
        os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("GSX_BENCH_HW_QUEUES", "8")
    import torch
    dist = None
    if world > 1 or os.environ.get("GSX_BENCH_FORCE_DIST"):   # (the variable: one-rank rehearsal of the N > 1 code path)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        # The control collectives of this script (id broadcast, self-check vote, barrier, max of the elapsed times) are host
        # values and go through gloo; device tensors (the torch.distributed fallback gather) go through nccl = RCCL, whose
        # communicator torch creates on first use only.  The pixels themselves are exchanged by gsx_gather on the library's
        # own RCCL communicator.  (A torch NCCL communicator that exists beside the library's streams costs every frame
        # ~0.03 ms on one GPU - measured with GSX_BENCH_PG=nccl - whether or not it is ever used.)
        pg = os.environ.get("GSX_BENCH_PG", "mixed")
        if pg == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            try:
                dist.init_process_group(backend="cpu:gloo,cuda:nccl")
            except Exception as e:      # noqa: BLE001 — no gloo on this host: everything through nccl, as before
                _dbg(f"gloo + nccl process group failed ({e}); nccl only")
                pg = "nccl"
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    _dbg(f"process group ready: rank {rank}/{world}")

    pkg = load_pkg()
    synth = importlib.import_module("gaussian_splatting_amd.synth")
    scene = importlib.import_module("gaussian_splatting_amd.scene")
    mg = importlib.import_module("gaussian_splatting_amd.multigpu")
    raw = synth.SCENES[scene_name](a.splats)
    splats = scene.raw_to_splats(pkg, raw, center=False)      # host half of ResourceManager (libgsx, C++)
    views = orbit_views(pkg, 64, W, H)
    # Poses differ a lot in cost on this scene (0.41 .. 0.74 ms over the first 8), which would show up
    # as a scaling loss that has nothing to do with N.  Weak scaling wants the per-GPU work fixed, so by default every rank
    # renders the pose of the N = 1 run; --distinct-views gives rank r pose r of the set instead.
    my_view = views[mg.shard_views(len(views), rank, max(world, 1))[0] if (a.distinct_views and not bands_mode) else 0]
    # the headline walks the 120-frame orbit (3 degrees per step): every rank the same poses by default (fixed per-GPU work),
    # --distinct-views: rank r starts 120 / N frames further along
    path = orbit_path(pkg, 120, W, H)
    path_unis = [pkg.make_uniforms(p[1], p[0]) for p in path]
    pose_off = (rank * (120 // max(world, 1))) if (a.distinct_views and not bands_mode) else 0
    g = pkg.Gsx(local_rank)
    g.upload(splats)
    for o in a.opt:
        k, v = o.split("=")
        g.set_option(k, int(v))
    # Sampled sorts (each one drains the two-frame pipeline, about one frame time): two of three bracket the sort's launches one
    # by one, the third keys + sort as a whole, every steps/4-th sort (8..16); in a short run every 5th sort, three of four by
    # launches (20 steps: sorts 0, 5, 10 = 6 launches, sort 15 as a whole).  Set again behind the warm-up so that the timed region's first sort
    # is a sampled one (the pipeline is empty there anyway).
    sample_every = int(os.environ.get("GSX_BENCH_SAMPLE_EVERY", 5 if a.steps < 40 else min(16, max(8, a.steps // 4))))
    g.set_option("sample_whole_every", int(os.environ.get("GSX_BENCH_SAMPLE_WHOLE_EVERY", 4 if a.steps < 40 else 3)))
    g.set_option("sample_every", sample_every)
    g.set_profiling(2)          # sampled sorts: every HBM pass of the sort (two of three), or keys + sort as a whole (the third), carry HIP events
    u_mine = pkg.make_uniforms(my_view[1], my_view[0])
    fb_bytes = W * H * 4
    bands = mg.shard_bands(H, max(world, 1))
    y0, y1 = bands[rank] if bands_mode else (0, H)

    # ---- the exchange: gsx_gather (RCCL point-to-point on the library's exchange stream), checked once before it is trusted ----
    gather_kind, gather_note = "none", None
    if dist is not None and a.gather != "none":
        gather_kind = a.gather
        if a.gather == "gsx":
            # every rank takes the same path through the collectives below whatever fails locally: rank 0 broadcasts the id
            # together with a "have one" byte, and nobody enters the (collective) communicator set-up without it
            ctl_dev = dev if pg == "nccl" else torch.device("cpu")
            idt = torch.zeros(129, dtype=torch.uint8, device=ctl_dev)
            if rank == 0 and world > 1:
                try:
                    idt[:128].copy_(torch.from_numpy(g.comm_unique_id()))
                    idt[128] = 1
                except Exception as e:      # noqa: BLE001 — librccl missing or unusable: fall back below
                    gather_note = f"gsx_comm_unique_id failed: {e}"
            if world > 1:
                dist.broadcast(idt, 0)
            try:
                if world > 1 and int(idt[128].item()) != 1:
                    raise RuntimeError("no RCCL id from rank 0")
                g.comm_init(idt[:128].cpu().numpy() if world > 1 else None, rank, world)
                probe = torch.full((4096,), rank + 1, dtype=torch.uint8, device=dev)
                sink = torch.zeros(4096 * world, dtype=torch.uint8, device=dev) if rank == 0 else None
                torch.cuda.synchronize()
                g.gather(probe.data_ptr(), 4096, sink.data_ptr() if rank == 0 else None,
                         [4096 * r for r in range(world)] if rank == 0 else None, [4096] * world if rank == 0 else None, 0)
                g.sync()
                ok = 1
                if rank == 0:
                    got = sink.cpu().numpy().reshape(world, 4096)
                    ok = int(all(np.all(got[r] == r + 1) for r in range(world)))
            except Exception as e:      # noqa: BLE001 — any failure here means: do not use this path
                ok, gather_note = 0, f"gsx_gather unavailable on rank {rank}: {e}"
            flag = torch.tensor([ok], dtype=torch.int32, device=ctl_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                gather_kind = "torch"
                gather_note = gather_note or "gsx_gather self-check failed on some rank; fell back to torch.distributed.gather"
    # the torch.distributed fallback moves device tensors: a nccl (= RCCL) group of its own, created only when it is used
    # (the decision above is the same on every rank)
    torch_group = None
    if dist is not None and gather_kind == "torch" and pg != "nccl":
        torch_group = dist.new_group(backend="nccl", device_id=dev)
    _dbg(f"gather: {gather_kind}")

    keep = []

    def alloc(nbytes):
        t = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        keep.append(t)
        return t.data_ptr(), t

    fg = None      # view batches through gsx_gather
    bg = None      # view batches through torch.distributed
    if a.gather_batch <= 0:
        a.gather_batch = 1 if gather_kind == "gsx" else 8
    if not bands_mode and gather_kind == "gsx":
        fg = mg.GsxFrameGather(g, rank, world, fb_bytes, a.gather_batch, alloc)
    elif not bands_mode and gather_kind == "torch":
        ext = torch.cuda.ExternalStream(g.stream(), device=dev)
        bg = mg.BatchedFrameGather(dist, rank, world, fb_bytes, a.gather_batch, dev, stream=ext, group=torch_group)
    band_recv = [torch.zeros(fb_bytes, dtype=torch.uint8, device=dev) for _ in range(world)] if (bands_mode and gather_kind == "torch" and rank == 0) else None
    frame_no = 0
    exchange = True     # switched off after the timed region: what follows is rank 0's alone
    band_pending = []   # bands mode: framebuffers of the frames whose exchange has not been posted yet
    band_fbs = [alloc(fb_bytes)[0] for _ in range(4)] if (bands_mode and gather_kind == "gsx") else []
    band_last = [None]  # framebuffer the last exchange of the timed region gathered into

    pose_no = 0
    last_pose = [None]
    rb_ring = [None]    # set by the read-back extra

    # Sort ahead (library option sort_ahead): the step queues the sort of the NEXT pose, then this pose's frame, which draws the
    # permutation its own pose was sorted for one step earlier.  Every step is still one sort and one frame of a new pose; what
    # changes is that the next sort reaches the device before the host has worked through this frame's ~30 launches instead of
    # after them (tools/host_vs_gpu.py: in call order sort, render, sort the device waited for the host between frames).
    sort_ahead = not a.in_order and not (bands_mode and a.band_cull)
    if sort_ahead:
        g.set_option("sort_ahead", 1)
    primed = [None]     # the pose whose permutation waits to be drawn

    def step(pose=None, uniforms=None):
        nonlocal frame_no, pose_no
        nxt = None
        if pose is None and not a.static_view:
            k = (pose_no + pose_off) % len(path)
            pose, uniforms = path[k][2], path_unis[k]
            pose_no += 1
            nxt = path[(pose_no + pose_off) % len(path)][2]
        pos = my_view[2] if pose is None else pose
        uni = u_mine if uniforms is None else uniforms
        last_pose[0] = (pos, uni)
        if sort_ahead:
            if primed[0] is not pos:                 # first step, or the caller left the orbit: forget what waits, sort this pose now
                g.set_option("sort_ahead", 1)
                g.sort(pos, download=False)
            primed[0] = pos if nxt is None else nxt
            g.sort(primed[0], download=False)        # the NEXT pose's sort, queued before this pose's frame
        else:
            if bands_mode and y1 > y0 and a.band_cull and world > 1:      # (one rank = one band = the whole frame: nothing to cull)
                g.select_band(uni, W, H, y0, y1, want_count=False)     # cull by band BEFORE the sort: on the sort stream, no host wait
            g.sort(pos, download=False)          # queued behind the running frame
        if bands_mode:
            if exchange and gather_kind == "gsx":
                # Frames go round a ring of 4 framebuffers and the exchange of frame k is posted two frames late, just before
                # frame k+2 is queued: gsx_gather completes the frame it ships, and that is the one wait for frame k the
                # library's two-frames-in-flight pipeline has at this point anyway; frame k+2 renders into other memory, so it
                # does not wait for the exchange either.
                if len(band_pending) == 2:
                    mg.gather_bands_gsx(g, band_pending.pop(0), bands, W, rank)
                g.set_framebuffer_dev(band_fbs[frame_no % len(band_fbs)], fb_bytes)
            if y1 > y0:
                g.render(uni, W, H, y0, y1, download=False)
            if not exchange:
                pass
            elif gather_kind == "gsx":
                band_pending.append(band_fbs[frame_no % len(band_fbs)])
            elif gather_kind == "torch":
                g.sync()
                fb_ptr = g.framebuffer_dev()[0]
                t = torch.as_tensor(DevArray(fb_ptr, fb_bytes), device=dev)
                mg.gather_to_rank0(dist, t, rank, world, band_recv, group=torch_group)
        else:
            batcher = fg or bg
            if batcher is not None:
                slot = batcher.slot(frame_no)
                if fg is not None:
                    g.set_framebuffer_dev(slot, fb_bytes)              # (the frames already queued keep their own slots)
                else:
                    g.set_framebuffer_dev(slot.data_ptr(), fb_bytes)
                if frame_no:
                    if bg is not None:
                        g.stream()                                     # (orders the main stream, where torch gathers, behind the frames queued so far)
                    batcher.frame_done(frame_no - 1)                   # closes a batch every gather_batch frames
            if rb_ring[0] is not None:      # (extras only: every frame read back, deferred, into a ring of page-locked frames)
                g.render(uni, W, H, out=rb_ring[0][frame_no % 3])
            else:
                g.render(uni, W, H, download=False)
        frame_no += 1

    def drain():
        """Complete the last frame and send what is left of the current batch."""
        batcher = fg or bg
        while band_pending:
            band_last[0] = band_pending.pop(0)
            mg.gather_bands_gsx(g, band_last[0], bands, W, rank)
        if bands_mode:
            g.set_framebuffer_dev(None)
        if batcher is not None and frame_no:
            g.set_framebuffer_dev(None)
            if bg is not None:
                g.stream()
            batcher.frame_done(frame_no - 1)
            batcher.finish()

    def fence():
        """Everything this rank has queued is complete, then a barrier (an all-reduce of a host word over gloo), then once more."""
        g.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.all_reduce(torch.zeros(1, dtype=torch.int32, device=dev if pg == "nccl" else "cpu"))
        g.sync()
        torch.cuda.synchronize()

    _dbg("scene uploaded")
    for i in range(a.warmup):
        step()
    fence()
    _dbg("warm-up done")
    # a full collection of Python's cyclic garbage collector walks every object torch and numpy have created (tens of
    # milliseconds, seen as ONE stalled step some hundred steps in): collect now and park what exists in the permanent generation
    import gc
    gc.collect()
    gc.freeze()
    g.scatter_samples(reset=True)
    g.sort_samples(reset=True)
    g.set_option("sample_every", sample_every)           # (restarts the count: the first timed sort is sampled)
    # Frames are queued back to back (the next frame's sort is queued behind the running frame; gsx_render waits for the
    # previous frame only to check whether it needs more iterations).  Nothing is read back inside the loop: the sampled
    # launches sit in the library's event ring and are collected after the closing fence.
    stamps = [] if os.environ.get("GSX_BENCH_STEPTIMES") else None    # (diagnostic: host time after every step's calls)
    calls = []
    if stamps is not None:                               # (diagnostic: host time inside every library call of the loop)
        def _timed(name, fn):
            def w(*x, **kw):
                t = time.perf_counter()
                r = fn(*x, **kw)
                calls.append((name, t - t0, time.perf_counter() - t))
                return r
            return w
        g.sort, g.render = _timed("sort", g.sort), _timed("render", g.render)
    t0 = time.perf_counter()
    for i in range(a.steps):
        step()
        if stamps is not None:
            stamps.append(time.perf_counter() - t0)
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    _dbg("timed region closed")
    if stamps is not None and rank == 0:
        np.save(os.environ["GSX_BENCH_STEPTIMES"], np.asarray(stamps))
        with open(os.environ["GSX_BENCH_STEPTIMES"] + ".calls.txt", "w") as fh:
            for name, at, dur in calls:
                fh.write(f"{at * 1e3:9.3f} ms  {name:6s} {dur * 1e6:8.1f} us\n")
    per_pass, pass_n = g.scatter_samples(reset=True)     # mean launch duration of each radix pass over the timed region
    sort_ms_timed, sort_n_timed = g.sort_samples(reset=True)
    t = g.timings()
    pairs, iters, passes = t["tile_pairs"], t["render_iters"], t["sort_passes"]
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if pg == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- bands: rank 0's gathered frame against the frame it renders alone (frames are a pure function of scene,
    #      permutation and uniforms, so the composition must be byte-identical) ----
    bands_match = None
    if bands_mode and rank == 0:
        g.sync()
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        gathered = np.empty(fb_bytes, np.uint8)
        if gather_kind == "torch":
            gathered = mg.compose_bands(band_recv, bands, W).cpu().numpy()
        else:
            hip.hipMemcpy(C.c_void_p(gathered.ctypes.data), C.c_void_p(band_last[0] or g.framebuffer_dev()[0]), C.c_size_t(fb_bytes), C.c_int(2))
        g.select_all()
        g.set_option("sort_ahead", 0)                    # (forgets the permutation sorted ahead for a frame that never came)
        primed[0] = None
        g.sort(last_pose[0][0], download=False)
        alone = g.render(last_pose[0][1], W, H).reshape(-1)
        bands_match = bool(np.array_equal(gathered, alone)) if (world == 1 or gather_kind != "none") else None
    # ---- sort-ahead: the last frame of the timed region against the same pose rendered in call order (sort, then render) ----
    ahead_match = None
    if sort_ahead and rank == 0 and not bands_mode and a.steps > 0:
        import ctypes as C
        g.sync()
        hip = C.CDLL("libamdhip64.so")
        ptr = g.framebuffer_dev()[0]
        timed_last = np.empty(fb_bytes, np.uint8)
        hip.hipMemcpy(C.c_void_p(timed_last.ctypes.data), C.c_void_p(ptr), C.c_size_t(fb_bytes), C.c_int(2))
        g.set_framebuffer_dev(None)
        g.set_option("sort_ahead", 0)                    # (forgets the permutation sorted ahead for a frame that never came)
        primed[0] = None
        g.sort(last_pose[0][0], download=False)
        ahead_match = bool(np.array_equal(timed_last, g.render(last_pose[0][1], W, H).reshape(-1)))
    fg = bg = None                                       # everything below is rank 0's alone: no collective
    exchange = False

    # ---- beside the headline, N = 1 only: no history, and a moving camera ----
    extras = {}
    acc, n_prof = {}, 6
    if rank == 0 and not a.no_extras:
        g.set_framebuffer_dev(None)
        if world == 1 and not bands_mode:
            K2 = max(10, min(a.steps, 40))
            for i in range(3):
                step(my_view[2], u_mine)
            g.sync()
            t1 = time.perf_counter()
            for i in range(K2):
                step(my_view[2], u_mine)
            g.sync()
            extras["ms_per_step_static"] = (time.perf_counter() - t1) / K2 * 1e3
            extras["static_note"] = ("the SAME pose every step: the history-driven schedule is perfectly predicted and every tile is binned only up to the "
                                     "rank at which it saturated in the identical previous frame (not the headline: that replays knowledge of the answer)")
            g.set_option("no_history", 1)
            for i in range(3):
                step(my_view[2], u_mine)
            g.sync()
            t1 = time.perf_counter()
            for i in range(K2):
                step(my_view[2], u_mine)
            g.sync()
            extras["ms_per_step_no_history"] = (time.perf_counter() - t1) / K2 * 1e3
            # what the instrumentation costs: the same two figures with no sort sampled (a sampled sort waits for both frame slots
            # to drain so that its events bracket the sort alone, then the pipeline fills again)
            g.set_profiling(0)
            for i in range(3):
                step(my_view[2], u_mine)
            g.sync()
            t1 = time.perf_counter()
            for i in range(K2):
                step(my_view[2], u_mine)
            g.sync()
            extras["ms_per_step_no_history_unsampled"] = (time.perf_counter() - t1) / K2 * 1e3
            g.set_option("no_history", 0)
            for i in range(3):
                step()
            g.sync()
            t1 = time.perf_counter()
            for i in range(a.steps):
                step()
            drain()
            g.sync()
            extras["ms_per_step_unsampled"] = (time.perf_counter() - t1) / max(a.steps, 1) * 1e3
            extras["unsampled_note"] = ("the headline's workload (a new pose every step, continuing along the orbit) and the no-history figure with "
                                        "profiling level 0: no sort of the loop carries events; `value` keeps the instrumented time")
            # the PCIe-inclusive figure: the same loop with EVERY frame read back into host memory (8.3 MB at 1080p), by the
            # deferred read-back (the pixels of call k arrive while frame k + 1 runs) into a ring of three page-locked frames
            if fg is None and bg is None:
                g.set_option("deferred_readback", 1)
                rb_ring[0] = [pkg.pinned_frame(H, W) for _ in range(3)]
                for i in range(3):
                    step()
                g.sync()
                t1 = time.perf_counter()
                for i in range(a.steps):
                    step()
                g.sync()
                extras["ms_per_step_with_readback"] = (time.perf_counter() - t1) / max(a.steps, 1) * 1e3
                extras["readback_note"] = ("the unsampled loop with every frame copied to host memory (PCIe-inclusive; option deferred_readback, "
                                           "page-locked ring of three frames); never `value`: the headline's frames stay in HBM")
                rb_ring[0] = None
                g.set_option("deferred_readback", 0)
            g.set_profiling(2)
        # stage breakdown on a few extra frames with every stage bracketed (outside the timed region)
        g.set_profiling(1)
        for i in range(n_prof + 1):
            step()
            tm = g.timings()
            if i:                                  # first frame after switching levels is a warm-up
                for k, val in tm.items():
                    acc[k] = acc.get(k, 0.0) + val

    if rank == 0:
        K, n = a.steps, a.splats
        frames_total = K * (1 if bands_mode else world)
        n_pass = int(round(passes))
        form = int(t.get("sort_form", 1))
        # the roofline kernels: every sampled launch priced at its own bytes
        if form == 2:
            stage_bytes = {0: PARTITION_BYTES, 1: LOCAL_BYTES}
            launch_bytes = {p: stage_bytes.get(p, LOCAL_BYTES) * n for p in per_pass}
        else:   # (pass 0 of the radix form generates the index: 12 B per key)
            launch_bytes = {p: ((SCATTER_BYTES_PASS0 if p == 0 else SCATTER_BYTES_PASSN) - (4 if (n_pass < 4 and p == n_pass - 1) else 0)) * n
                            for p in per_pass}      # (the last ungated pass stores no keys)
        sum_ms = sum(per_pass[p] * pass_n[p] for p in per_pass)
        sum_bytes = sum(launch_bytes[p] * pass_n[p] for p in per_pass)
        launches = sum(pass_n.values())
        launch_ms = sum_ms / launches if launches else 0.0
        achieved = sum_bytes / (sum_ms * 1e-3) / 1e9 if sum_ms > 0 else 0.0
        sort_bytes = sort_bytes_per_splat(n_pass, form=form)
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "r3_sort_traffic.json" if form == 2 else "r2_scatter_traffic.json")     # PMC passes (tools/pmc_traffic.sh), committed
        if os.path.exists(tf):
            rec = json.load(open(tf))
            if rec.get("splats") == n:
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_src = f"profiles/{os.path.basename(tf)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/pmc_traffic.sh), not measured in this run"
        out = {
            "metric": f"splats_sorted_per_s (whole {W}x{H} frame: keys+sort+project+bin+raster)",
            "value": n * frames_total / elapsed,
            "unit": "splats/s",
            "n_gpus": world,
            "steps": K,
            "warmup": a.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if bands_mode else "weak",
            "vs_baseline": None,
            "dtype": "f32 keys / u32 radix",
            "data": "synthetic",
            "config": {"workload": (f"cfg5: {n}-splat overlap-stress scene, ONE {W}x{H} frame per step in {world} tile band(s)" if bands_mode else
                                    f"cfg2: {n}-splat clustered synthetic scene, single view {W}x{H} per step" +
                                    (", the same pose every step" if a.static_view else ", a new camera pose every step (120-frame orbit, 3 degrees apart)")),
                       "splats": n, "width": W, "height": H, "views_per_step_per_gpu": None if bands_mode else 1,
                       "poses": ("one pose" if a.static_view else "120-frame orbit, one new pose per step") +
                                ("" if bands_mode else ("; rank r starts 120/N frames further along" if a.distinct_views else "; every rank renders the N=1 poses (fixed per-GPU work)")),
                       "sharding": ("screen-space tile bands over replicated splats" if bands_mode else "camera views over replicated splats"),
                       "gather": {"gsx": f"gsx_gather (grouped RCCL send/recv on the library's exchange stream)" + ("" if bands_mode else f", batches of {a.gather_batch} frames"),
                                  "torch": "torch.distributed.gather (nccl = RCCL)", "none": "none"}[gather_kind]},
            "fps": frames_total / elapsed,
            "tile_pairs": int(pairs),
            "render_iters": iters,
            "sort_passes": n_pass,
            "async_sort": not any(o.startswith("async_sort=0") for o in a.opt),
            "sort_form": "two-pass (keys + top-digit counts, column scan, partition, in-LDS bucket sort)" if form == 2 else "LSD radix passes",
            "roofline": {"kernel": ("k2_partition + k2_local (the two HBM passes of the depth sort behind the key kernel)" if form == 2 else
                                    "k_radix_scatter<u32> (one stable radix scatter pass of the depth sort)"), "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": launch_ms, "per_pass_launch_ms": [per_pass[p] for p in sorted(per_pass)],
                         "per_pass_bytes": [launch_bytes[p] for p in sorted(per_pass)],
                         "launches_per_step": n_pass, "launches_timed": int(launches),
                         "algorithmic_bytes_per_launch": sum_bytes / launches if launches else None,
                         "note": "HIP events inside the timed region: two of three sampled sorts (one sort in 5, or in steps/4 = 8..16, is sampled) bracket each of these launches; the sampled sorts run "
                                 "on the main stream, un-overlapped (the others run beside the previous frame's rasterizer, where the same launch takes about twice as long: "
                                 "DESIGN.md 4.6); " + ("partition: 4 B key in + 8 B pair out, bucket sort (its two size classes + the gated overflow kernel as one bracket): 8 B in + 4 B out"
                                                       if form == 2 else "pass 0 priced at 12 B/key, passes >= 1 at 16 B/key")},
        }
        if not bands_mode:
            out["fps_1080p"] = out["fps"]
        if gather_note:
            out["config"]["gather_note"] = gather_note
        if dist is not None:
            out["config"]["control_collectives"] = ("torch.distributed over nccl" if pg == "nccl" else
                                                    "torch.distributed over gloo (host words: id broadcast, barrier, max of the elapsed times); pixels over RCCL")
        if bands_match is not None:
            out["bands_match_single_gpu_frame"] = bands_match
        out["call_order"] = ("sort k+1, render k (library option sort_ahead: every frame draws the permutation sorted for its own pose one step earlier)"
                             if sort_ahead else "sort k, render k")
        if ahead_match is not None:
            out["sort_ahead_last_frame_matches_in_order"] = ahead_match
        if sort_n_timed:
            out["sort_only_ms"] = sort_ms_timed
            out["sort_only_samples"] = sort_n_timed
            out["sort_only_note"] = "keys + sort bracketed as a whole by HIP events in every third sampled sort INSIDE the timed region (a sampled sort: on the main stream, not overlapped)"
            out["sort_only_splats_per_s"] = n / (sort_ms_timed * 1e-3)
            out["sort_only_bytes_per_splat"] = sort_bytes
            out["sort_only_hbm_frac"] = sort_bytes * n / (sort_ms_timed * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["sort_only_hbm_frac_at_92B"] = 92 * n / (sort_ms_timed * 1e-3) / 1e9 / HBM_PEAK_GBS
            # the same figures inside the roofline object: per launch the two-pass form moves fewer bytes per unit of on-chip work than
            # an LSD pass did, so `frac` of a launch went DOWN while the sort got faster; the whole sort at SURVEY 8(d)'s price compares
            out["roofline"]["whole_sort"] = {"ms": sort_ms_timed, "samples": sort_n_timed, "bytes_per_splat_survey_8d": 92,
                                             "achieved": 92 * n / (sort_ms_timed * 1e-3) / 1e9, "unit": "GB/s",
                                             "frac": 92 * n / (sort_ms_timed * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "bytes_per_splat_this_design": sort_bytes,
                                             "frac_at_this_design_bytes": sort_bytes * n / (sort_ms_timed * 1e-3) / 1e9 / HBM_PEAK_GBS}
        out.update(extras)
        if acc:
            stage = {k: acc[k] / n_prof for k in ("select_ms", "keygen_ms", "sort_ms", "scatter_ms", "project_ms", "bin_ms", "raster_ms", "total_ms")}
            out["stage_ms"] = stage
            out["stage_ms_note"] = f"{n_prof} extra frames after the timed region with every stage bracketed by HIP events (adds ~65 us per frame)"
            shades, blends = acc["raster_shades"] / n_prof, acc["raster_blends"] / n_prof
            staged = acc.get("raster_staged", 0.0) / n_prof
            r_ms = stage["raster_ms"]
            if r_ms > 0:
                out["roofline_raster"] = {
                    "kernel": "k_raster (tile rasterizer, all iterations of a frame)", "bound": "valu (packed f32) + lds",
                    "raster_ms": r_ms, "pairs_binned": int(pairs), "pairs_staged": staged, "wave_splat_shades": shades, "pixel_blends": blends,
                    "blends_per_s": blends / (r_ms * 1e-3),
                    "lane_hit_rate": blends / (shades * 128.0) if shades else None,
                    "valu_tflops": blends * BLEND_FLOP / (r_ms * 1e-3) / 1e12,
                    "valu_tflops_issued": shades * 128 * BLEND_FLOP / (r_ms * 1e-3) / 1e12,
                    "valu_peak_tflops": VALU_PEAK_TFLOPS,
                    "valu_frac_issued": shades * 128 * BLEND_FLOP / (r_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                    "lds_gbs": shades * SHADE_LDS_BYTES / (r_ms * 1e-3) / 1e9, "lds_peak_gbs": LDS_PEAK_GBS,
                    "note": f"{BLEND_FLOP} flop per pixel-splat blend (+1 v_exp_f32); 'issued' counts all 128 pixels of a wave per visited splat (masked lanes still occupy the SIMD); "
                            f"LDS: {SHADE_LDS_BYTES} B of broadcast record reads per (wave, splat) visit; counters from the {n_prof} profiled frames"}
        kr = kernel_rooflines_from_profile(n) if not bands_mode else None
        if kr:
            out["kernel_rooflines"] = kr
        if a.cpu_frames > 0 and world == 1:                    # (the CPU baseline is an N = 1 figure: the other ranks would only wait for it)
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
    if dist is not None:      # (a host word over gloo, like every other control collective: no NCCL communicator is created just to leave)
        if pg == "nccl":
            dist.barrier()
        else:
            dist.all_reduce(torch.zeros(1, dtype=torch.int32))
    g.close()
    if dist is not None:
        dist.destroy_process_group()


class DevArray:
    """Zero-copy torch view of a raw device pointer (framebuffer) via __cuda_array_interface__."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": (nbytes,), "typestr": "|u1", "version": 2}


if __name__ == "__main__":
    main()
