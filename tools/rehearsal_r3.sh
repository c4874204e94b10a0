#!/bin/bash
# Round 3: what one GPU can rehearse of the N > 1 paths and of cfg 4, plus the sort harness (run on the GPU box from the repo root)
mkdir -p gpurun_out/r3
# N > 1 code path of bench.py with world size 1 (RCCL communicator of one rank, gsx_gather into itself)
GSX_BENCH_FORCE_DIST=1 timeout -k 10 200 python bench.py --cpu-frames 0 --steps 100 --no-extras > gpurun_out/r3/bench_forced_dist_n1.json 2> gpurun_out/r3/bench_forced_dist_n1.err || exit 1
GSX_BENCH_FORCE_DIST=1 timeout -k 10 200 python bench.py --cpu-frames 0 --steps 100 --no-extras --shard bands > gpurun_out/r3/bench_forced_dist_n1_bands.json 2> gpurun_out/r3/bench_forced_dist_n1_bands.err || exit 1
# cfg 4's shape: 20 M splats, a different pose of the 64-pose set per step, one rank
GSX_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --cpu-frames 0 --steps 64 --splats 20000000 --distinct-views --no-extras > gpurun_out/r3/bench_cfg4_n1.json 2> gpurun_out/r3/bench_cfg4_n1.err || exit 1
# the sort harness: every kernel of the two-pass sort bracketed by events, checked against std::sort on the host
# (arguments: splats scene reps [top-digit bits, 0 = the plan's] [camera inside the scene box] [third k2_local instance launched])
for args in "6000000 clustered 20" "6000000 uniform 20" "6000000 tight 20" "1000000 uniform 20" "20000000 clustered 10 12" "20000000 clustered 10" "20000000 clustered 10 0 0 1" "20000000 uniform 10 0 0 1" "20000000 clustered 10 0 1 1" "12000000 clustered 10 0 0 1" "6000000 clustered 20 0 1"; do
  echo "== tools/sort2bench $args"
  timeout -k 10 120 tools/sort2bench $args || exit 1
done > gpurun_out/r3/sort2bench.txt 2>&1
timeout -k 10 120 tools/rocprim_calib > gpurun_out/r3/rocprim_calib.jsonl 2>&1 || true
