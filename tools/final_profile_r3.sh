#!/bin/bash
# Round-end evidence: the default bench line, then the same command under rocprofv3 --kernel-trace --stats.
set -e
timeout -k 10 300 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
# the command the driver runs at round end
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_driver_cmd.json 2> gpurun_out/r3_bench_driver_cmd.err
timeout -k 10 300 python bench.py --steps 120 --warmup 10 --cpu-frames 0 > gpurun_out/r3_bench_120.json 2> gpurun_out/r3_bench_120.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r3f
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3f -o r3f -- python3 bench.py --cpu-frames 0 > gpurun_out/r3_bench_under_rocprof.json 2> gpurun_out/r3_bench_under_rocprof.err
python tools/kstats.py gpurun_out/prof_r3f > gpurun_out/r3_bench_kernel_stats.txt
cp gpurun_out/prof_r3f/*/*kernel_stats.csv gpurun_out/r3_bench_kernel_stats.csv 2>/dev/null || cp gpurun_out/prof_r3f/*kernel_stats.csv gpurun_out/r3_bench_kernel_stats.csv
python tools/ktrace.py gpurun_out/prof_r3f 48 0.5 > gpurun_out/r3_bench_kernel_trace_excerpt.txt
cat gpurun_out/r3_bench_kernel_stats.txt
python -c "
import json; d=json.load(open('gpurun_out/r3_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline'])"
# the same command with everything on one stream, one frame at a time: per-kernel durations that are the kernels' own
rm -rf gpurun_out/prof_r3s
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3s -o r3s -- python3 bench.py --cpu-frames 0 --in-order --opt async_sort=0 --opt frames_in_flight=1 > gpurun_out/r3_bench_serialized_under_rocprof.json 2> gpurun_out/r3_bench_serialized_under_rocprof.err
python tools/kstats.py gpurun_out/prof_r3s > gpurun_out/r3_bench_serialized_kernel_stats.txt
cp gpurun_out/prof_r3s/*/*kernel_stats.csv gpurun_out/r3_bench_serialized_kernel_stats.csv 2>/dev/null || cp gpurun_out/prof_r3s/*kernel_stats.csv gpurun_out/r3_bench_serialized_kernel_stats.csv
python tools/chunk_count_roofline.py gpurun_out/prof_r3s > gpurun_out/r3_chunk_count_roofline.json
python tools/ktrace.py gpurun_out/prof_r3s 48 0.2 > gpurun_out/r3_bench_serialized_kernel_trace_excerpt.txt
cat gpurun_out/r3_bench_serialized_kernel_stats.txt gpurun_out/r3_chunk_count_roofline.json
