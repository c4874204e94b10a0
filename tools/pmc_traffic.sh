#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters, one counter per rocprofv3 pass (run on the GPU box):
#   tools/pmc_traffic.sh <outdir>
# Calibrates FETCH_SIZE / WRITE_SIZE on a 4-byte-per-lane streaming copy of known size (the scatter kernel's
# access width; MI355X_MICROARCH.md: widths other than 16 B per lane are uncalibrated), then collects both
# counters for bench.py's kernels.  tools/pmc_traffic.py turns the csv files into profiles/r2_scatter_traffic.json.
set -e
out=$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 60 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/cal4_$ctr -- tools/membw 96 copy4 > $out/cal4_$ctr.log 2>&1
  timeout -s KILL 60 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/cal16_$ctr -- tools/membw 96 copy16 > $out/cal16_$ctr.log 2>&1
  timeout -s KILL 170 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/bench_$ctr -- python3 bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-extras > $out/bench_$ctr.log 2>&1
done
