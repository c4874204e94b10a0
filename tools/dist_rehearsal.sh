#!/bin/bash
# One-rank rehearsal of bench.py's N > 1 code path on a single GPU (RCCL gather with world size 1).
for extra in "" "--no-gather" "--gather-batch 1"; do
  GSX_BENCH_FORCE_DIST=1 timeout -k 10 200 python -X faulthandler bench.py --cpu-frames 0 --steps 100 $extra > gpurun_out/bench_dist1.json 2> gpurun_out/bench_dist1.err
  echo "exit=$? [$extra]"
  grep -i "error\|traceback" gpurun_out/bench_dist1.err | tail -5
  python -c "
import json; d=json.load(open('gpurun_out/bench_dist1.json')); print(d['ms_per_step'], d['config']['sharding'], d['tile_pairs'], d['render_iters'])" || true
done
