#!/bin/bash
# HBM traffic of the LOD cut's three launches at cfg 3 from PMC counters, one counter per rocprofv3 pass (run on the GPU box):
#   tools/pmc_lod_r3.sh <outdir>        then   python tools/pmc_lod_r3.py <outdir>
# Same recipe as tools/pmc_traffic.sh: FETCH_SIZE / WRITE_SIZE calibrated on streaming copies of known size.
set -e
out=$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GSX_OPTS="async_sort=0 frames_in_flight=1"
export GSX_CFG3_FRAMES=24
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 60 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/cal4_$ctr -- tools/membw 96 copy4 > $out/cal4_$ctr.log 2>&1
  timeout -s KILL 60 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/cal16_$ctr -- tools/membw 96 copy16 > $out/cal16_$ctr.log 2>&1
  timeout -s KILL 170 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/cfg3_$ctr -- python3 tools/configs_table.py cfg3 > $out/cfg3_$ctr.log 2>&1
done
