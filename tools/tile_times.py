"""Debug experiment (needs a -DGSX_DEBUG_TILETIME build of libgsx, GSX_LIB=tools/_dbg/libgsx_dbg.so): when every retired tile
of the rasterizer started and ended (100 MHz wall clock, parked in the tile's unused accumulators) - is the launch bound by
throughput or by its longest tiles?"""
import ctypes as C, sys, os, importlib
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from conftest import load_pkg
pkg = load_pkg()
synth = importlib.import_module("gaussian_splatting_amd.synth")
scene = importlib.import_module("gaussian_splatting_amd.scene")
cam = importlib.import_module("gaussian_splatting_amd.camera")
import numpy as np
W, H = 1920, 1080
NT = (W // 16) * ((H + 15) // 16)
for name in sys.argv[1:] or ("clustered",):
    raw = synth.SCENES[name](6_000_000); sp = scene.raw_to_splats(pkg, raw)
    g = pkg.Gsx(0); g.upload(sp)
    g.set_option("frames_in_flight", 1); g.set_option("async_sort", 0)
    w = cam.orbit_pose(5.0, []); v, p, pos = cam.camera_matrices(w, 45.0, W / H)
    for rep in range(4):
        g.sort(pos, download=False)
        g.render(pkg.make_uniforms(p, v), W, H, download=False); g.sync()
        t = g.timings()
        acc = np.zeros(NT * 256 * 4, np.float32)
        st = np.zeros(NT + 2, np.uint32)
        assert g.lib.gsx_debug_read(g.ctx, 13, acc.ctypes.data_as(C.c_void_p), C.c_uint64(acc.nbytes)) == 0
        assert g.lib.gsx_debug_read(g.ctx, 10, st.ctypes.data_as(C.c_void_p), C.c_uint64(st.nbytes)) == 0
        tt = acc.view(np.uint64).reshape(NT, 512)[:, :2]
        ret = st[:NT] == 2
        t0, t1 = tt[ret, 0].astype(np.int64), tt[ret, 1].astype(np.int64)
        dur = (t1 - t0) / 100.0
        span = (t1.max() - t0.min()) / 100.0
        order = np.argsort(t1)
        print(name, "rep", rep, "pairs", t["tile_pairs"], "iters", t["render_iters"], "retired", int(ret.sum()), "of", NT,
              "| span us %.1f" % span, "| tile us: mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()),
              "| busy tile-us / span %.1f" % (dur.sum() / span))
        # how many tiles are still running in the last stretch of the launch
        end = t1.max()
        for back in (100, 50, 25, 10):
            at = end - back * 100
            print("    %3d us before the end: %4d tiles running, %4d not started" % (back, int(((t0 <= at) & (t1 > at)).sum()), int((t0 > at).sum())))
        # k_chunk_emit: one record per 512-pair slice, 8 MB into the accumulator array
        sl = acc.view(np.uint64)[(1 << 20):(1 << 20) + 4 * 65536].reshape(-1, 4)
        n_sl = (int(t["tile_pairs"]) + 511) // 512 if t["render_iters"] == 1 else 0
        if n_sl:
            sl = sl[:n_sl]
            ok = (sl[:, 3] == 1) & (np.arange(n_sl) % 128 != 0)      # (every 128th record shares its line with a tile's stamps)
            s0, s1 = sl[ok, 0].astype(np.int64), sl[ok, 1].astype(np.int64)
            sd = (s1 - s0) / 100.0
            vis, blks = (sl[ok, 2] & 0xFFFFFFFF).astype(np.int64), (sl[ok, 2] >> 32).astype(np.int64)
            print("    emit: %d slices, span us %.1f | slice us: mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f | visits mean %.1f max %d | blocks mean %.1f max %d"
                  % (ok.sum(), (s1.max() - s0.min()) / 100.0, sd.mean(), *np.percentile(sd, [50, 90, 99]), sd.max(), vis.mean(), vis.max(), blks.mean(), blks.max()))
            idx = np.nonzero(ok)[0]
            slow = np.argsort(-sd)[:12]
            print("      slowest slices (index, us, start offset us, visits, blocks):", [(int(idx[i]), round(float(sd[i]), 1), round(float((s0[i] - s0.min()) / 100.0), 1), int(vis[i]), int(blks[i])) for i in slow])
            late = np.argsort(-s0)[:6]
            print("      latest starts (index, start offset us, us):", [(int(idx[i]), round(float((s0[i] - s0.min()) / 100.0), 1), round(float(sd[i]), 1)) for i in late])
            hist, edges = np.histogram(sd, bins=[0, 4, 8, 12, 16, 24, 32, 48, 64, 1000])
            print("      slice us histogram:", list(zip([int(e) for e in edges[:-1]], [int(h) for h in hist])))
            for lo, hi in ((0, 0.1), (0.1, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 0.9), (0.9, 1.0)):
                m = (idx >= lo * n_sl) & (idx < hi * n_sl)
                print("      slices %3d%%-%3d%%: us mean %.2f max %.2f, visits mean %.1f, blocks mean %.1f, start offset us mean %.1f" % (lo * 100, hi * 100, sd[m].mean(), sd[m].max(), vis[m].mean(), blks[m].mean(), ((s0[m] - s0.min()) / 100.0).mean()))
        long = np.argsort(-dur)[:5]
        print("    longest:", [(int(np.nonzero(ret)[0][i]), round(float(dur[i]), 1), round(float((t0[i] - t0.min()) / 100.0), 1)) for i in long], "(tile, us, start offset us)")
    g.close()
